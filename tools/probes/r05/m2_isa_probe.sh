#!/bin/bash
# tools/m2_isa_probe.cpp over the code objects tools/m2_isa_patch.py made (tools/bin/m2_<patch>.co): one update of a full chip by
# the in-place form <8 x 4 components, 8 groups> against the two-operand-set form of the same code object
cd "$(dirname "$0")/../../.." || exit 1
T=_ZN6fluhip18nmf_update5_kernelILi8ELi8ELi6ELi1ELi0ELi2ELi1ELi0ELi0EEEvNS_8Upd5ArgsE
R=_ZN6fluhip18nmf_update5_kernelILi8ELi8ELi6ELi1ELi0ELi1ELi1ELi0ELi0EEEvNS_8Upd5ArgsE
for co in ${*:-tools/bin/m2_*.co}; do
  for steps in 1 35; do
    echo "== $co steps $steps"
    timeout 120 tools/bin/m2_isa_probe $co $T $R $steps 3 2>&1 | tail -5
  done
done
