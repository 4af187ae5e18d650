#!/bin/bash
# Dev: a VARIANT of libtensoflow_hip.so with extra -D switches on some translation units (instrumented / ablation builds: -DBVH_STATS,
# -DBVH_CLOCK, -DTF_DEV, -DTF_ABLATE_DMA ...), linked with the product's other objects.  Never loaded by the product or the tests: the
# tools/exp_* scripts point tensoflow_amd.lib.LIB_PATH at it.
#   tools/build_variant.sh <name> "<extra hipcc flags>" <file.hip> [<file.hip> ...]    ->  build_variants/lib_<name>.so
set -e
name=$1; flags=$2; shift 2
root=$(cd "$(dirname "$0")/.." && pwd)
src=$root/tensoflow_amd/csrc
out=$root/build_variants
mkdir -p $out/obj_$name
make -s -C $src >/dev/null          # the product's objects are what the variant links against
objs=""
for f in $src/*.o; do
  b=$(basename $f .o); skip=0
  for v in "$@"; do [ "$(basename $v .hip)" = "$b" ] && skip=1; done
  [ $skip = 0 ] && objs="$objs $f"
done
for v in "$@"; do
  b=$(basename $v .hip)
  extra=""; [ "$b" = "view_angles" ] && extra="-fno-slp-vectorize"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value $extra $flags -c $src/$b.hip -o $out/obj_$name/$b.o
  objs="$objs $out/obj_$name/$b.o"
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs -o $out/lib_$name.so
echo "$out/lib_$name.so"
