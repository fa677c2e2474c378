#!/bin/bash
# A/B by an environment switch of the library, interleaved on one box:  tools/ab_env.sh VAR CONFIG MODE
var=$1; cfg=$2; mode=$3
for rep in 1 2 3; do
  echo -n "default      "; python3 tools/run_config.py $cfg 20 3 $mode | sed 's/.*| wall/| wall/'
  echo -n "$var=1 "; env $var=1 python3 tools/run_config.py $cfg 20 3 $mode | sed 's/.*| wall/| wall/'
done
