timeout 300 python tools/ab_compare.py --config cfg3 --blends 128 --steps 20 2>&1 | tail -10
python tools/stage_cycles.py 2>&1 | tail -8 | head -7
timeout 600 python bench.py --steps 100 --warmup 10 --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg3', d['value'], d['roofline']['phases_ms'])"
timeout 600 python bench.py --config cfg1 --steps 100 --warmup 10 --no-cpu | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('cfg1', d['value'], d['roofline']['phases_ms'])"
timeout 1500 python -m pytest tests -x -q -m gpu -k "fused or fft or hsc or convolution or per_band or tiny or forward or synthetic or random" 2>&1 | tail -3
