#!/bin/bash
# Dev tool: build an alternative libtensoflow_hip.so with one source recompiled under extra -D flags.
#   tools/build_variant.sh <name> <source.hip> [-DFOO=1 ...]   ->   build_variants/lib_<name>.so
set -e
cd "$(dirname "$0")/.."
name=$1; src=$2; shift 2
mkdir -p build_variants
make -C tensoflow_amd/csrc -j8 >/dev/null
obj=build_variants/${name}_${src%.hip}.o
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value "$@" -c tensoflow_amd/csrc/$src -o $obj
others=$(ls tensoflow_amd/csrc/*.o | grep -v "/${src%.hip}.o$")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $obj $others -o build_variants/lib_${name}.so
echo build_variants/lib_${name}.so
