#!/bin/bash
# Builds a variant of the kernel library next to the product one, for A/B runs on the GPU box:
#   tools/dev/build_variant.sh epi1 -DAOT_LEAN_EPI=1      ->  aot-benchmark_amd/csrc/libaot_hip_epi1.so
# (no GPU needed; same per-source flags as the product build; the variant libraries are git-ignored and travel with the gpurun
#  snapshot like the product library; AOT_HIP_LIB=<path> makes any script or test load one)
set -e
name=$1; shift
cd "$(dirname "$0")/../../aot-benchmark_amd/csrc"
python - "$name" "$@" <<'PY'
import sys
import build
print(build.build_lib(verbose=False, variant=sys.argv[1], defines=sys.argv[2:]))
PY
ls -la libaot_hip_$name.so
