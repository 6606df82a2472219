for i in 1 2; do
  for lib in new base; do
    if [ $lib = base ]; then export D2T_PROBE_LIB=doc2tex_amd/csrc/libd2t_base.so; else unset D2T_PROBE_LIB; fi
    python3 bench.py --gpus 1 --steps 24 --warmup 6 --no-cpu-baseline --no-secondary 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'])" || exit 1
  done
done
