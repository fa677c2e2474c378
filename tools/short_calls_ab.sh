#!/bin/bash
# short march calls, planner's choice at 1 M walls: default against a switch of the library (tools/short_calls.py)
var=${1:-HEAT_AMD_NO_ZIGZAG}
for rep in 1 2; do
python3 tools/short_calls.py | grep "1000000 walls, planner"
echo "--- $var=1"
env $var=1 python3 tools/short_calls.py | grep "1000000 walls, planner"
done
