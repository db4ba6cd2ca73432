#!/bin/bash
# Runs ON THE GPU BOX: the seeded sweeps on other cases (SSMI_FUZZ_SALT = $1..$2), one pytest process per file and salt,
# each under its own time limit; stops at the first failure (no step is tried again).  Logs: gpurun_out/soak<salt>_<file>.log
P="python -u -m pytest -x -q -m gpu --timeout 120 --timeout-method thread"
for salt in $(seq ${1:-1} ${2:-1}); do
  export SSMI_FUZZ_SALT=$salt
  for f in fuzz_gpu fuzz_api_gpu fuzz_manager_gpu fuzz_files_gpu multi_device_gpu zz_no_alloc; do
    timeout -k 10 300 $P tests/test_$f.py > gpurun_out/soak${salt}_$f.log 2>&1 || { echo "salt $salt $f FAILED"; tail -n 40 gpurun_out/soak${salt}_$f.log; exit 1; }
    echo "salt $salt $f: $(tail -n 1 gpurun_out/soak${salt}_$f.log)"
  done
done
