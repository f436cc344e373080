# Sweeps POSEHIP_PAF_SUB (ranges per limb in paf_score_kernel) inside the bench step; run on the GPU box from the repo root.
for n in 2 3 4 6 8 4 2; do echo "# POSEHIP_PAF_SUB=$n"; POSEHIP_PAF_SUB=$n BENCH_NO_H2D_LEG=1 timeout -k 10 100 python bench.py --cpu-frames 0 --steps 60 --sustained-steps 0 2>/dev/null | python3 -c "
import json,sys;d=json.loads(sys.stdin.read().strip().splitlines()[-1]);print(d['value'],d['roofline']['decode_ms_per_step'])"; done
