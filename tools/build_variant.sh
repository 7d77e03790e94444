#!/bin/bash
# A/B builds of one translation unit: tools/build_variant.sh TAG [-DMACRO=... ...]
# compiles VARIANT_SRC (default bdrt_nuts_k0.hip: the headline instantiations of bdrt_nuts16.h; bdrt_nuts.hip: the one-chain kernels)
# with the extra flags and links it with the objects of the default build into bayes_drt_amd/variants/libbdrt_TAG.so;
# run with BDRT_LIBRARY=bayes_drt_amd/variants/libbdrt_TAG.so.
set -e
TAG=$1; shift
SRC=${VARIANT_SRC:-bdrt_nuts_k0.hip}
BASE=${SRC%.hip}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/bayes_drt_amd/csrc"
mkdir -p ../variants
ILP=""; case "$BASE" in bdrt_nuts_k0|bdrt_nuts_k1) ILP="-mllvm -amdgpu-sched-strategy=max-ilp";; esac
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -w -mllvm -disable-machine-licm $ILP "$@" -c $SRC -o ../variants/${BASE}_$TAG.o
OBJS=$(ls *.o | grep -v "^$BASE.o$")
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/libbdrt_$TAG.so ../variants/${BASE}_$TAG.o $OBJS
echo built bayes_drt_amd/variants/libbdrt_$TAG.so
