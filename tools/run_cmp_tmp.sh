set -e
timeout -k 10 400 python -u -m pytest tests/test_mfma_gpu.py tests/test_flat_gpu.py tests/test_ivf_gpu.py -m gpu -q -x --timeout 200 --timeout-method thread > gpurun_out/t_mfma.log 2>&1 || { tail -30 gpurun_out/t_mfma.log; exit 1; }
tail -2 gpurun_out/t_mfma.log
for B in 1024 4096; do
  timeout -k 10 200 python bench.py --storage f16 --batch $B --steps 10 --warmup 2 --no-extra --no-ivf-extra --no-cpu-baseline > gpurun_out/new_b$B.json 2> gpurun_out/new_b$B.err
  python -c "import json;j=json.load(open('gpurun_out/new_b$B.json'));print('B',$B,round(j['roofline']['kernel_avg_ms'],3),round(j['roofline']['frac'],4))"
done
timeout -k 10 200 python bench.py --workload ivf --steps 20 --warmup 3 --no-cpu-baseline > gpurun_out/new_ivf.json 2> gpurun_out/new_ivf.err; python -c "
import json;j=json.load(open('gpurun_out/new_ivf.json'));print({k:v for k,v in j.get('extra',{}).items() if 'qps' in k}, j['value'])"
export SSMI_LIB=$PWD/semantic-search_amd/csrc/libssmi_stamp.so
for B in 1024 4096; do timeout -k 10 200 python bench.py --storage f16 --batch $B --steps 3 --warmup 1 --no-extra --no-ivf-extra --no-cpu-baseline > gpurun_out/stamp_b$B.json 2> gpurun_out/stamp_b$B.err; grep stamp gpurun_out/stamp_b$B.err | tail -2; done
