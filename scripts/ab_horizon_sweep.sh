for lib in build_ab/pre_fusion.so build_ab/fused_v9.so build_ab/pre_fusion.so build_ab/fused_v9.so; do echo $lib; CEM_MPC_LIB=$PWD/$lib python scripts/time_rollout_vs_horizon.py | python -c "
import sys,json
rows=[json.loads(l) for l in sys.stdin]
for r in rows:
    if r['H'] in (1,8,30,60): print(r['segments'], r['H'], round(r['rollout_us'],1), end=' | ')
print()
"; done
