#!/bin/bash
# Round 5, GPU session V: the generic measuring pass keeps every string's length (k_mplp_len_x) and the emit goes straight to its writing walk;
# STA_GENERIC_PASSES=2 = the emit measuring for itself (session I's form), =1 = round 4's per-string walks
cd $GRAFT_REPO_ROOT
O=gpurun_out/r05v; mkdir -p $O
( time timeout 600 python -m pytest tests/test_gpu_synth.py tests/test_gpu_goldens.py tests/test_kputd.py tests/test_gpu_cabi_client.py -q -o timeout=240 -p no:cacheprovider -k "generic_walker or mpileup_sO or output_extra or 79 or kputd or mods or mp2" ) > $O/pytest_generic.log 2>&1; tail -3 $O/pytest_generic.log
( time timeout 600 python -m pytest tests/test_gpu_benchsize_parity.py -q -o timeout=400 -p no:cacheprovider -k "sOx" ) > $O/pytest_sox.log 2>&1; tail -3 $O/pytest_sox.log
run() { env $1 python bench.py --steps 8 --warmup 2 --workload mpileup30_B_sOx --no-cpu-baseline --no-pmc --no-e2e 2>/dev/null | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$1', round(d['ms_per_step'],3), round(d['value']), {k: round(x,3) for k,x in list(d['kernels_ms_per_step'].items())[:3]}, d['output_sha256'][:10])"; }
for v in STA_GENERIC_PASSES=2 STA_GENERIC_PASSES=0 STA_GENERIC_PASSES=2 STA_GENERIC_PASSES=0; do run $v; done 2>&1 | tee $O/ab_generic.log
timeout 500 python scripts/hunt4.py 4 5 > $O/hunt4.log 2>&1; tail -1 $O/hunt4.log
true
