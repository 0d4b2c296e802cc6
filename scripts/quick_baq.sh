cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_synth.py -m gpu -q -x -k "baq or EA" 2>&1 | tail -1
python bench.py --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(round(d['value']), round(d['ms_per_step'],2), {k: round(v,2) for k,v in list(d['kernels_ms_per_step'].items())[:6]})"
