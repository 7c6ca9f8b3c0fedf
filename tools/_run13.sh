cd /root/repo
for d in 0 16 4 20; do echo "DBG=$d"; DVS_WINO_KSPLIT=0 DVS_WINO_DBG=$d timeout -k 10 300 python tools/wino_fixed_cost.py 12 2>&1 | grep shape | python3 -c "
import sys, json
for l in sys.stdin:
    d = json.loads(l); print('  ', d['shape'], 'fixed/wg', d['fixed_us_per_wg'], 'kstep/wg', d['us_per_kstep_per_wg'], d['points'])"; done
