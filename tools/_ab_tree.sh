cd "$GRAFT_REPO_ROOT"
timeout 200 python -m pytest tests -m gpu -x -q -k "fused_heads" 2>&1 | tail -2
one() { timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-stream-figure 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('hot_path_ms'))"; }
for i in 1 2 3; do echo -n "new: "; one; done
