#!/bin/bash
# Device assembly of every unit, searched for instructions this library must not contain:
#   v_ashr_pk_u8_i32 / v_ashr_pk_i8_i32 -- hipcc (ROCm 7.2) ORs their result as if bits 31:16 were zero; gfx950 leaves those bits as the
#   destination register held them (a wrong blue in every third pixel of the RGB expand until its clamp was spelled out).
# bash tools/check_isa.sh   (CPU only: hipcc cross-compiles; ~2 minutes)
cd "$(dirname "$0")/../pixlzr-rust_amd/csrc" || exit 2
bad=0
for u in pxz_expand pxz_shrink32 pxz_shrink64 pxz_shrink_generic pxz_oklab pxz_stream pxz_frames pxz_tree; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -S --cuda-device-only -o /tmp/check_isa_$u.s $u.hip 2>/dev/null || { echo "$u: does not compile"; bad=1; continue; }
  n=$(grep -c 'v_ashr_pk_u8_i32\|v_ashr_pk_i8_i32' /tmp/check_isa_$u.s)
  echo "$u: $n"
  [ "$n" -ne 0 ] && bad=1
  rm -f /tmp/check_isa_$u.s
done
exit $bad
