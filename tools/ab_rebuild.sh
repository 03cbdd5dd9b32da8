#!/bin/bash
# same-box A/B of the whole rebuild: the last commit (tools/build_head_variant.sh) against the working tree, 3 rounds
for i in 1 2 3; do
  echo -n "head: "; LBVH_LIB=build_exp/liblbvh_head.so python tools/build_only.py 2>&1 | grep "rebuild(" | tr '\n' ' '; echo
  echo -n "work: "; python tools/build_only.py 2>&1 | grep "rebuild(" | tr '\n' ' '; echo
done
