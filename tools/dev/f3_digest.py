import json,sys
for f in sys.argv[1:]:
    d=json.loads(open(f).read().strip().splitlines()[-1])
    ps=d['per_step']
    print(f, round(d['value']), round(d['wall_s'],2), d.get('complete_rate'), d.get('collision_rate'), 'late steps:', [(r['running'], r['solve_ms']) for r in ps[-12::3]])
