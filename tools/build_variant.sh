#!/bin/bash
# A/B or diagnostic build of the library: tools/build_variant.sh NAME "-DMI_CONV_TIMELINE" [files...]
# recompiles the named translation units (default: the five conv_igemm*.hip) with the extra flags into
# yolov7_d2_amd/csrc/_build_NAME/ and links them with the regular objects into yolov7_d2_amd/libmi355det_NAME.so
set -e
cd "$(dirname "$0")/../yolov7_d2_amd/csrc"
NAME=$1; FLAGS=$2; shift 2 || true
FILES=${@:-conv_igemm conv_igemm_kc16 conv_igemm_kc32 conv_igemm_kc64 conv_igemm_kc128}
make -j8 >/dev/null
mkdir -p _build_$NAME
CXX="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -Wno-unused-value -Wno-unused-result -Wno-inline-asm"
for f in $FILES; do /opt/rocm/bin/hipcc $CXX $FLAGS -c $f.hip -o _build_$NAME/$f.o & done; wait
OBJS=""
for o in *.o; do b=${o%.o}; if [ -f _build_$NAME/$o ]; then OBJS="$OBJS _build_$NAME/$o"; else OBJS="$OBJS $o"; fi; done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmi355det_$NAME.so $OBJS
echo built yolov7_d2_amd/libmi355det_$NAME.so
