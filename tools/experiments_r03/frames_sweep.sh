mkdir -p gpurun_out/sweep
for c in 3 2; do for f in 8 16 20 24 32 48 64; do
 python bench.py --config $c --frames $f --no-cpu-baseline --no-verify --no-host-abi 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg',$c,'frames',$f,'ms',d['ms_per_step'],'us/frame',round(1000*d['ms_per_step']/$f,3),d['roofline']['kernel'][:32])"
done; done | tee gpurun_out/sweep/frames.txt
