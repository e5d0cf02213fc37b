for w in 8 4 2; do timeout -k 10 300 python tools/shard_step.py --world $w 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-900; done
