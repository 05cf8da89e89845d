for lib in $(ls $PWD/gpurun_scratch/libtex_*.so | sort -t_ -k2 -n); do
  NYXHIP_LIB=$lib python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 --gray-depth 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$(basename $lib)', 'ms', round(d['ms_per_step'],3))"
done
python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-check --no-extras --tile-path-tiles 0 --gray-depth 64 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('full ms', round(d['ms_per_step'],3))"
