#!/bin/bash
# diagnostic builds (select one with SSMI_LIB=<path>):
#   libssmi_stamp.so     cycle stamps per row tile in the 256-query MFMA kernel (prints [stamp] lines)
#   libssmi_ivfstamp.so  wall-clock stamps per wave in the per-query IVF scan (SSMI_IVF_STAMP_PRINT=1 prints [ivf stamp])
#   libssmi_scanstamp.so wall-clock stamp per wave at the end of its rows in K1 (SSMI_SCAN_STAMP_PRINT=1 prints [scan stamp])
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DNDEBUG"
/opt/rocm/bin/hipcc $F -DSSMI_STAMP -c mfma_pp.hip -o /tmp/mfma_pp_stamp.o &   # stamps per row tile only: s_memtime inside the K loop doubles the kernel time
/opt/rocm/bin/hipcc $F -DSSMI_STAMP -c mfma.hip -o /tmp/mfma_stamp.o &
/opt/rocm/bin/hipcc $F -DSSMI_STAMP -c mfma_n128.hip -o /tmp/mfma_n128_stamp.o &
/opt/rocm/bin/hipcc $F -DSSMI_STAMP -c mfma_r128.hip -o /tmp/mfma_r128_stamp.o &
/opt/rocm/bin/hipcc $F -DSSMI_IVF_STAMP -c ivf.hip -o /tmp/ivf_stamp.o &
/opt/rocm/bin/hipcc $F -DSSMI_SCAN_STAMP -c scan_f32.hip -o /tmp/scan_f32_stamp.o &
/opt/rocm/bin/hipcc $F -DSSMI_SCAN_STAMP -c scan_f16.hip -o /tmp/scan_f16_stamp.o &
/opt/rocm/bin/hipcc $F -DSSMI_SCAN_STAMP -c api.hip -o /tmp/api_stamp.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libssmi_stamp.so $(ls build/*.o | grep -v 'build/mfma') /tmp/mfma_stamp.o /tmp/mfma_pp_stamp.o /tmp/mfma_n128_stamp.o /tmp/mfma_r128_stamp.o -Wl,-rpath,/opt/rocm/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libssmi_ivfstamp.so $(ls build/*.o | grep -v 'build/ivf.o') /tmp/ivf_stamp.o -Wl,-rpath,/opt/rocm/lib
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libssmi_scanstamp.so $(ls build/*.o | grep -v -E 'build/(scan_f32|scan_f16|api)\.o') /tmp/scan_f32_stamp.o /tmp/scan_f16_stamp.o /tmp/api_stamp.o -Wl,-rpath,/opt/rocm/lib
