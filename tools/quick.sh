#!/bin/bash
# Quick timing pass over the bench workloads (GPU box, repo root): tools/quick.sh OUT_PREFIX
# headline planned + streamed, config 3, config 5, partitions: tools/run_config.py lines.
out=${1:-gpurun_out/quick}
mkdir -p $(dirname $out)
for spec in "headline 20 5 plan" "headline 20 3 stream" "3 20 3 stream" "5 20 3 stream" "partitions 20 5 plan" "2 20 20 plan"; do
    python3 tools/run_config.py $spec >> $out.txt 2>> $out.err || echo "FAILED: $spec" >> $out.txt
done
cat $out.txt
