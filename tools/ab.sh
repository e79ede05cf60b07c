#!/bin/bash
# developer aid (run on the GPU box): kernel time of the configs[2] scan for several builds of the library / settings
mkdir -p gpurun_out
for v in "$@"; do
  case $v in
    base) L=$PWD/rac-2d_amd/libracgpu.so; E="";;
    w*) L=$PWD/rac-2d_amd/libracgpu.so; E="RACGPU_WAVES_PER_CU=${v#w}";;
    *) L=$PWD/build/libvar_$v.so; E="";;
  esac
  env $E RACGPU_LIB=$L timeout -k 10 200 python tools/dev/dev_scan_grid.py rate06_dipole_reformated_again_withgrain_lowH2Bind.dat 1 > gpurun_out/ab_$v.log 2>&1
  echo "$v: $(grep -m1 kernel gpurun_out/ab_$v.log | cut -c1-60) | $(grep -m1 'cycles per step' gpurun_out/ab_$v.log)"
done
