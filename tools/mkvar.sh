#!/bin/bash
# developer aid: build/libvar_<name>.so = the library with extra -D flags (tools/ab.sh runs them side by side)
# usage: tools/mkvar.sh name -DRG_X=1 ...
set -e
name=$1; shift
mkdir -p build/var_$name
cd rac-2d_amd/csrc
F="--offload-arch=gfx950 -ffp-contract=off -munsafe-fp-atomics -mllvm -disable-machine-licm"
hipcc -O3 -fPIC -std=c++17 -I../../include $F "$@" -c engine.hip -o ../../build/var_$name/engine.o
hipcc -shared -fPIC -o ../../build/libvar_$name.so ../../build/var_$name/engine.o network.o hc_host.o multi.o -ldl -lpthread
echo built build/libvar_$name.so
