#!/bin/bash
# In the build container: scratch copy of the reference's Python package (unmodified) under oracle/_ref/refpkg (git-ignored; it travels
# to the GPU box with the snapshot), then scripts/gpu_reference_package.py on the MI355X.  Evidence only: nothing in tests/, bench.py or
# smoke() reads oracle/_ref/refpkg.
set -e
cd "$(dirname "$0")/.."
rm -rf oracle/_ref/refpkg && mkdir -p oracle/_ref/refpkg
cp -r /root/reference/python-package/gpboost oracle/_ref/refpkg/gpboost
/usr/local/graft/bin/gpurun --timeout 600 -- 'mkdir -p gpurun_out/refpkg; timeout 500 python scripts/gpu_reference_package.py > gpurun_out/refpkg/reference_package_on_mi355x.log 2>&1; tail -25 gpurun_out/refpkg/reference_package_on_mi355x.log'
