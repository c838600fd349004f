#!/bin/bash
# builds an A/B variant of the library: bash profiles/dev/build_variant.sh NAME [extra hipcc flags]  ->  mapad_amd/variant_NAME.so (use with MAPAD_AMD_LIB)
NAME=$1; shift
cd "$(dirname "$0")/../.."
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -ffp-contract=off -fno-fast-math -Xarch_device -O2 -fno-builtin-log2f -fno-builtin-powf -fno-builtin-expf -fno-builtin-exp2f -fno-builtin-log10f -mllvm -disable-promote-alloca-to-lds -w "$@" -o mapad_amd/variant_$NAME.so mapad_amd/csrc/mapad_amd.hip mapad_amd/csrc/index_gpu.hip && echo built mapad_amd/variant_$NAME.so
