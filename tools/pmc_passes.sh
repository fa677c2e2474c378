#!/bin/bash
# rocprofv3 counter passes of one command, each set of counters in a run of its own (never together with a trace):
#   tools/pmc_passes.sh OUT_DIR -- python3 tools/run_config.py 3 20 1 stream
# writes OUT_DIR/{insts,f64,cycles,fetch,write}/.../*counter_collection.csv
set -e
out=$1; shift; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $out/insts -- "$@" > $out/insts.log 2>&1
rocprofv3 --pmc SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $out/f64 -- "$@" > $out/f64.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES --output-format csv -d $out/cycles -- "$@" > $out/cycles.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -- "$@" > $out/fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -- "$@" > $out/write.log 2>&1
echo "pmc passes done: $out"
