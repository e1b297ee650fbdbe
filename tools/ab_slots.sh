# development aid: N=1 bench under different slot / stream settings
for cfg in "16 2" "24 2" "24 3" "32 2" "32 4"; do
  set -- $cfg
  echo "== slots=$1 work_streams=$2"
  SPP_MAX_SLOTS=$1 SPP_WORK_STREAMS=$2 timeout -k 10 200 python bench.py --slots $1 --no-cpu-baseline --no-model-step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']/1e9,3), round(d['roofline']['avg_launch_ms'],4))"
done
