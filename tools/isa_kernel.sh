#!/bin/bash
# Device assembly of ONE instantiation of the 16-chain NUTS kernel (bdrt_nuts16.h): tools/isa_kernel.sh NJ MODE TA [outdir]
# -> <outdir>/nuts_kernel_NJ_MODE_TA.s (kernel body only) and .blocks (per-basic-block instruction mix, tools/isa_blocks.py)
set -e
NJ=${1:-11}; MODE=${2:-2}; TA=${3:-1}; OUT=${4:-/tmp/isa}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
mkdir -p "$OUT"
F="$OUT/probe_${NJ}_${MODE}_${TA}.hip"
printf '#include "bdrt_nuts16.h"\nnamespace bdrt { BDRT_NUTS16_DEFINE(%s, %s, %s) }\n' "$NJ" "$MODE" "$TA" > "$F"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp -w \
    -I"$ROOT/bayes_drt_amd/csrc" $ISA_FLAGS --offload-device-only -S "$F" -o "$OUT/probe_${NJ}_${MODE}_${TA}.s"
S="$OUT/nuts_kernel_${NJ}_${MODE}_${TA}.s"
awk '/^_ZN4bdrt11nuts_kernel.*:/{p=1} p{print} /^\.Lfunc_end/{if(p)exit}' "$OUT/probe_${NJ}_${MODE}_${TA}.s" > "$S"
grep -E "^; (NumVgprs|ScratchSize|Occupancy|NumSgprs)" "$OUT/probe_${NJ}_${MODE}_${TA}.s" | head -4
python3 "$ROOT/tools/isa_blocks.py" "$S" > "${S%.s}.blocks"
tail -1 "${S%.s}.blocks"
