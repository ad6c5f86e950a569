#!/bin/bash
# tools/build_variant.sh NAME "-DFLAG ..." ["unit.hip.o ..."] -- a second build of the library for an A/B on one box (tools/gpu_ab_lib.sh): the
# named translation units (default: those that hold the gate kernel, kernels.hip.o kernels_ops.hip.o) are recompiled with the extra flags, every
# other object is taken from acvm_amd/build, and the result is linked to tools/ab/libacvm_amd_NAME.so (git-ignored; load it with ACVM_AMD_LIB=...).
set -e
NAME=$1; FLAGS=$2; UNITS=${3:-"kernels.hip.o kernels_ops.hip.o"}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
python -m acvm_amd.build > /dev/null
mkdir -p "$ROOT/tools/ab/obj_$NAME"
OBJS=""
for o in "$ROOT"/acvm_amd/build/*.o; do
  b=$(basename "$o")
  case " $UNITS " in
    *" $b "*)
      /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -x hip -Wall -Wno-unused-result -Wno-unused-value -ffp-contract=off $FLAGS \
        -c "$ROOT/acvm_amd/csrc/${b%.o}" -o "$ROOT/tools/ab/obj_$NAME/$b" &
      OBJS="$OBJS $ROOT/tools/ab/obj_$NAME/$b";;
    *) OBJS="$OBJS $o";;
  esac
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o "$ROOT/tools/ab/libacvm_amd_$NAME.so" $OBJS -lz
echo "built tools/ab/libacvm_amd_$NAME.so"
