#!/bin/bash
# A/B library with some XCD-fused kernel instances rebuilt under extra -D flags (everything else is taken from the shipped build):
#   tools/ab_build.sh <name> "<instance ids | unit names>" "<-D flags>"   ->  webgpu-fft_amd/lib_e<name>/libmi355fft.so   (select with MI355FFT_LIB=...)
# an entry that is not a number names a translation unit of csrc/ (lines_fam_row_big, mixed_ct_kernels, ...)
set -e
cd "$(dirname "$0")/../webgpu-fft_amd/csrc"
name=$1; ids=$2; defs=$3
mkdir -p ../build_e$name ../lib_e$name
objs=""
for f in ../build/*.o; do
  b=$(basename $f .o); skip=0
  for i in $ids; do [ "$b" = "lines_xcd_$i" ] && skip=1; [ "$b" = "$i" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $f"
done
for i in $ids; do
  if [[ "$i" =~ ^[0-9]+$ ]]; then
    hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -I. -I../../include $defs -DMI355_XCD_ID=$i -c lines_xcd_one.hip -o ../build_e$name/lines_xcd_$i.o &
  else
    hipcc -std=c++17 -O3 --offload-arch=gfx950 -fPIC -fvisibility=hidden -Wall -Wno-unused-function -I. -I../../include $defs -c $i.hip -o ../build_e$name/$i.o &
  fi
done
wait
for i in $ids; do if [[ "$i" =~ ^[0-9]+$ ]]; then objs="$objs ../build_e$name/lines_xcd_$i.o"; else objs="$objs ../build_e$name/$i.o"; fi; done
hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib_e$name/libmi355fft.so $objs -Wl,--no-undefined
echo "built lib_e$name"
