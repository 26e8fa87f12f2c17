#!/bin/bash
# same-box A/B of harness builds: run_ab.sh name1 name2 ...   (each: build/native/<name> time <cfg>), two rounds
for round in 1 2; do for b in "$@"; do for cfg in "16 4096 5 1" "16 4096 5 2" "16 1024 10 2" "8 4096 5 2"; do echo -n "$b: "; timeout 120 build/native/$b time $cfg | sed 's/X3W_ABL=0 //; s/ | max.*//'; done; done; done
