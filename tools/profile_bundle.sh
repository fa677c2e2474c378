#!/bin/bash
# The evidence bundle of a round, on the GPU box (from the repo root):  tools/profile_bundle.sh r02
# For the headline (cluster-resident and streamed) and BASELINE configs 3 and 5: rocprofv3 --kernel-trace --stats of
# the bench command, and the --pmc passes (each in a run of its own) that bench.py's `roofline` reads back.
# Writes under gpurun_out/<tag>_bundle/ and copies the summaries into profiles/ (to be committed).
set -e
tag=$1
part=${2:-all}   # stats | counters | all (two gpurun calls when one is too long)
R=$GRAFT_REPO_ROOT
[ -z "$R" ] && R=$(pwd)
out=$R/gpurun_out/${tag}_bundle
mkdir -p $out
export TMPDIR=/tmp
stats() {  # name, bench args...
    local name=$1; shift
    (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace_$name -- python3 $R/bench.py "$@" --no-cpu-baseline --no-extras > $out/${name}_under_rocprof.json 2> $out/${name}_under_rocprof.err) || true
    cp $(ls $out/trace_$name/*/*kernel_stats.csv | head -n 1) $R/profiles/${tag}_${name}_kernel_stats.csv
    cp $out/${name}_under_rocprof.json $R/profiles/${tag}_${name}_under_rocprof.json
    echo "stats $name done"
}
if [ "$part" != counters ]; then
stats headline --steps 200 --warmup 20 --no-configs
stats cfg3 --config 3 --steps 100 --warmup 20
stats cfg5 --config 5 --steps 100 --warmup 20
stats cfg2 --config 2 --steps 200 --warmup 20
stats partitions --config partitions --steps 100 --warmup 20
stats buildings40 --config buildings40 --steps 100 --warmup 20
fi
[ "$part" = stats ] && { mkdir -p $out/profiles && cp $R/profiles/${tag}_* $out/profiles/; echo "bundle $tag stats done"; exit 0; }
# counters: tools/run_config.py CONFIG P REPS MODE
$R/tools/pmc_passes.sh $out/pmc_headline_fused -- python3 $R/tools/run_config.py headline 20 2 plan
python3 $R/tools/make_counters.py ${tag}_headline_fused $out/pmc_headline_fused "k_surfaces_fast<16, 0, 1, 0, 4" headline 1000000 32000000 fused 20
$R/tools/pmc_passes.sh $out/pmc_headline_streamed -- python3 $R/tools/run_config.py headline 10 1 stream
python3 $R/tools/make_counters.py ${tag}_headline_streamed $out/pmc_headline_streamed "k_surfaces_fast<16, 0, 1, 0, 0" headline 1000000 32000000 streamed 1
$R/tools/pmc_passes.sh $out/pmc_cfg3 -- python3 $R/tools/run_config.py 3 10 1 stream
python3 $R/tools/make_counters.py ${tag}_cfg3_streamed $out/pmc_cfg3 "k_surfaces_stream" 3 1000000 32609258 streamed 1
$R/tools/pmc_passes.sh $out/pmc_cfg5 -- python3 $R/tools/run_config.py 5 10 1 stream
python3 $R/tools/make_counters.py ${tag}_cfg5_streamed $out/pmc_cfg5 "k_surfaces_stream" 5 200000 2096503 streamed 1
$R/tools/pmc_passes.sh $out/pmc_b40 -- python3 $R/tools/run_config.py buildings40 20 2 plan
python3 $R/tools/make_counters.py ${tag}_buildings40_fused $out/pmc_b40 "k_surfaces_fast<16, 0, 1, 0, 4, 2" buildings40 999840 31994880 fused 20
$R/tools/pmc_passes.sh $out/pmc_partitions -- python3 $R/tools/run_config.py partitions 20 2 plan
python3 $R/tools/make_counters.py ${tag}_partitions_fused $out/pmc_partitions "k_surfaces_fast<16, 0, 1, 0, 4" partitions 999936 31997952 fused 20
$R/tools/pmc_passes.sh $out/pmc_cfg2 -- python3 $R/tools/run_config.py 2 20 5 plan
python3 $R/tools/make_counters.py ${tag}_cfg2_fused $out/pmc_cfg2 "k_surfaces_fast<16, 0, 1, 0, 4" 2 10000 200000 fused 20
# where a cluster-resident workgroup's time goes (diagnostic build with in-kernel stamps)
python3 $R/tools/fused_phases.py headline 20 > $R/profiles/${tag}_fused_phases.txt
python3 $R/tools/fused_phases.py headline 5 >> $R/profiles/${tag}_fused_phases.txt
python3 $R/tools/fused_phases.py partitions 20 >> $R/profiles/${tag}_fused_phases.txt
mkdir -p $out/profiles && cp $R/profiles/${tag}_* $out/profiles/   # (gpurun brings back gpurun_out/ only)
echo "bundle $tag done"
