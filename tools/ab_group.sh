# development aid: N=1 bench under different group sizes
for cfg in "16 8" "32 16" "24 12" "20 10"; do
  set -- $cfg
  echo "== slots=$1 group=$2"
  SPP_MAX_SLOTS=$1 SPP_GROUP_SIZE=$2 timeout -k 10 200 python bench.py --slots $1 --no-cpu-baseline --no-model-step 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['ms_per_step'],4), round(d['value']/1e9,3), round(d['roofline']['avg_launch_ms'],4))"
done
