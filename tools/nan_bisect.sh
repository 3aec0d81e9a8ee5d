#!/bin/bash
# tools/nan_bisect.sh "ENV1=v ENV2=v" ...   -- per configuration: 4 processes x 800 eager steps at B = 2, counts the processes that hit a non-finite step
for cfg in "$@"; do
  bad=0
  for i in 1 2 3; do
    out=$(env $cfg timeout -k 10 200 python tools/nan_loop.py 2 700 2>&1 | tail -1)
    case "$out" in *"first non-finite"*) bad=$((bad+1));; esac
  done
  echo "config [$cfg]: $bad of 3 processes hit a non-finite step"
done
