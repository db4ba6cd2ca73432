#!/bin/bash
# diagnostic build: libssmi_stamp.so = libssmi.so with cycle stamps in the MFMA kernel (SSMI_LIB selects it)
set -e
cd "$(dirname "$0")"
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-gpu-rdc -DNDEBUG -DSSMI_STAMP"  # stamps per row tile only: s_memtime inside the K loop doubles the kernel time
/opt/rocm/bin/hipcc $F -c mfma_pp.hip -o /tmp/mfma_pp_stamp.o &
/opt/rocm/bin/hipcc $F -c mfma.hip -o /tmp/mfma_stamp.o &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o libssmi_stamp.so $(ls build/*.o | grep -v 'build/mfma') /tmp/mfma_stamp.o /tmp/mfma_pp_stamp.o -Wl,-rpath,/opt/rocm/lib
