"""CPU: build hygiene of gpboost_amd/csrc -- every file a translation unit #includes (directly) is a prerequisite of its object in the
Makefile, so an edit can never leave a stale object behind (a missing gpb_tree.inc dependency once did exactly that)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "gpboost_amd", "csrc")


def _rules():
    txt = open(os.path.join(CSRC, "Makefile")).read().replace("\\\n", " ")
    rules = {}
    for m in re.finditer(r"^\$\(BUILD\)/([\w\$\(\)]+)\.o:\s*(.*)$", txt, flags=re.M):
        rules[m.group(1)] = m.group(2)
    return rules


def test_every_local_include_is_a_makefile_prerequisite():
    rules = _rules()
    units = {"gpb_hip": "gpb_hip.cpp", "gpb_c_api": "gpb_c_api.cpp", "gpb_optim": "gpb_optim.cpp", "nn_kernels": "nn_kernels.hip",
             "hist_kernels": "hist_kernels.hip", "dense_kernels": "dense_kernels.hip", "laplace_kernels": "laplace_kernels.hip",
             "leaf_kernels": "leaf_kernels.hip", "vecchia_aux_kernels": "vecchia_aux_kernels.hip", "vecchia_dispatch": "vecchia_kernels.hip"}
    for obj, src in units.items():
        assert obj in rules, obj
        prereq = rules[obj]
        todo, seen = [src], set()
        while todo:                                   # follow local includes transitively (.inc files are part of the unit)
            f = todo.pop()
            if f in seen:
                continue
            seen.add(f)
            for inc in re.findall(r'^\s*#include\s+"([^"]+)"', open(os.path.join(CSRC, f)).read(), flags=re.M):
                name = os.path.basename(inc)
                assert name in prereq, "%s includes %s but $(BUILD)/%s.o does not depend on it" % (f, inc, obj)
                if os.path.exists(os.path.join(CSRC, name)):
                    todo.append(name)
