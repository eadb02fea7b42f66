#!/bin/bash
# Counter passes for the bench (one rocprofv3 run per pass; counters only, no API tracing).
# usage (on the GPU box): bash profiles/pmc_passes.sh <outdir> [bench args...]
set -u
OUT=$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
ROOT=${GRAFT_REPO_ROOT:-/root/repo}
run() {
  name=$1; shift
  timeout -s KILL 240 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d "$OUT" -o "$name" -- python3 "$ROOT/bench.py" $BENCH_ARGS > "$OUT/$name.log" 2>&1 || return 1
}
BENCH_ARGS="--steps 1 --warmup 0 --no-cpu-baseline $*"
# the kernel sources these counters belong to (bench.py computes the same hash)
python3 -c "import sys; sys.path.insert(0, '$ROOT'); import bench; print(bench.kernel_source_hash())" > "$OUT/source_sha1.txt"
run sq1 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY &&
run sq2 SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_VMEM SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU_FMA_F32 SQ_ACTIVE_INST_ANY TA_BUSY_avr TA_TA_BUSY_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_PENDING_STALL_CYCLES_sum &&
run sq3 SQ_INST_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_CYCLES &&
run tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_EA0_RDREQ_sum GRBM_GUI_ACTIVE &&
run fetch FETCH_SIZE &&
run write WRITE_SIZE
echo "pmc passes exit $?"
