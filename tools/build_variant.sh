#!/bin/bash
# A/B builds of the headline sampler kernel: tools/build_variant.sh TAG [-DMACRO=... ...]
# compiles bdrt_nuts_k0.hip (the S1 instantiations of bdrt_nuts16.h) with the extra flags and links it with the objects of the
# default build into bayes_drt_amd/variants/libbdrt_TAG.so; run with BDRT_LIBRARY=bayes_drt_amd/variants/libbdrt_TAG.so.
set -e
TAG=$1; shift
ROOT=$(cd "$(dirname "$0")/.." && pwd)
cd "$ROOT/bayes_drt_amd/csrc"
mkdir -p ../variants
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=fast -w -mllvm -disable-machine-licm -mllvm -amdgpu-sched-strategy=max-ilp "$@" -c bdrt_nuts_k0.hip -o ../variants/k0_$TAG.o
OBJS=$(ls *.o | grep -v bdrt_nuts_k0.o)
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../variants/libbdrt_$TAG.so ../variants/k0_$TAG.o $OBJS
echo built bayes_drt_amd/variants/libbdrt_$TAG.so
