#!/bin/bash
# Memory-system view of one command's kernels (diagnostic; each counter set in a rocprofv3 run of its own, never with a trace):
#   tools/pmc_mem.sh OUT_DIR -- python3 tools/run_config.py 3 10 1 stream ; python3 tools/pmc_table.py OUT_DIR/*
# average latencies = *_LEVEL or *_LATENCY sums / request counts; TLB = UTCL1 misses / requests
set -e
out=$1; shift; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
pass() { local name=$1; shift; rocprofv3 --pmc "$@" --output-format csv -d $out/$name -- "${CMD[@]}" > $out/$name.log 2>&1 || echo "pass $name failed"; }
CMD=("$@")
pass vmem SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES
pass lds SQ_INSTS_LDS SQ_INST_LEVEL_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SMEM SQ_INST_LEVEL_SMEM
pass tcp1 TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum
pass tcp2 TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum
pass tlb TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_MULTI_MISS_sum
pass tcc1 TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum
pass tcc2 TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum
pass tcc3 TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_64B_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum
pass ta TA_BUSY_avr TA_TA_BUSY_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum
pass busy GRBM_GUI_ACTIVE TCC_BUSY_avr TCC_CYCLE_sum TD_TD_BUSY_sum TD_TC_STALL_sum
echo "pmc mem passes done: $out"
