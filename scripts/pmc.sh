#!/bin/bash
# PMC passes for the fused IS kernel (one rocprofv3 run per counter group; see MI355X_MICROARCH.md).
# usage (on the GPU box, from the repo root): bash scripts/pmc.sh <outdir-under-gpurun_out> <groups> [bench args...]
#   groups: comma list of fetch,write,sq1,grbm,sq2,sq3,lat
set -u
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; shift
GROUPS_=$1; shift
ARGS="${*:---grid 40 --steps 1 --warmup 1 --no-cpu-baseline}"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {
  name=$1; shift
  case ",$GROUPS_," in *",$name,"*) ;; *) return 0;; esac
  timeout -s KILL 170 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 $GRAFT_REPO_ROOT/bench.py $ARGS > "$OUT/$name.json" 2> "$OUT/$name.err" || { echo "pass $name failed"; tail -3 "$OUT/$name.err"; return 1; }
}
run fetch FETCH_SIZE &&
run write WRITE_SIZE &&
run sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_MFMA SQ_ACTIVE_INST_VALU &&
run grbm GRBM_GUI_ACTIVE GRBM_COUNT &&
# average latency of the L2's memory-side reads = RDREQ_LEVEL / RDREQ (L2 clocks): Infinity-Cache hits vs HBM misses
run lat TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum &&
run sq2 SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_LDS &&
run sq3 SQ_VALU_MFMA_COEXEC_CYCLES SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_LDS_UNALIGNED_STALL SQ_IFETCH
echo done
