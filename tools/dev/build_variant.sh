#!/bin/bash
# Builds a variant of the kernel library next to the product one, for A/B runs on the GPU box:
#   tools/dev/build_variant.sh epi1 -DAOT_LEAN_EPI=1      ->  aot-benchmark_amd/csrc/libaot_hip_epi1.so
# (no GPU needed; the variant libraries are git-ignored and travel with the gpurun snapshot like the product library)
set -e
name=$1; shift
cd "$(dirname "$0")/../../aot-benchmark_amd/csrc"
srcs=$(python - <<'PY'
import build
print(' '.join(build.SOURCES))
PY
)
flags=$(python - <<'PY'
import build
print(' '.join(build.FLAGS))
PY
)
/opt/rocm/bin/hipcc $flags "$@" $srcs -o libaot_hip_$name.so
ls -la libaot_hip_$name.so
