"""Ninth hunt (round 6): mate-overlap resolution (HTSlib tweak_overlap_quality) over pairs with rich CIGARs -- the generator of
tests/test_gpu_overlap_walk.py at 1 500 pairs per seed, 85 % of them with gap operations inside the overlap -- engine vs oracle through both
pairing lanes (partners staged by the input lane / the window's own name table) and under 50-read windows:
    STA_EXE=tests/cpu/hipemu/_build/plain/samtools_amd/bin/samtools-amd python scripts/hunt9.py <first seed> <one past the last>"""
import os, subprocess, sys
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(REPO, 'tests'))
from test_gpu_overlap_walk import write_overlap_sam
EXE = os.environ.get('STA_EXE', os.path.join(REPO, 'samtools_amd/bin/samtools-amd')); ORA = os.path.join(REPO, 'oracle/_build/oracle_samtools')
bad=0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    d='/tmp/hunt9_%d'%seed; os.makedirs(d, exist_ok=True)
    sam, fa = write_overlap_sam(d, seed, n_pairs=1500, n_ref=4000, plain_p=0.15)
    for args in (["mpileup","-B","-Q","0","-f",fa], ["mpileup","-f",fa]):
        want = subprocess.run([ORA]+args+[sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL).stdout
        for env in ({}, {"STA_OLAP_DEVICE_TABLE":"1"}, {"STA_WINDOW_READS":"50"}):
            got = subprocess.run([EXE]+args+[sam], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, env=dict(os.environ, **env)).stdout
            if got != want: bad+=1; print("DIFF", seed, args, env, flush=True)
print("hunt9: %d differences" % bad)
