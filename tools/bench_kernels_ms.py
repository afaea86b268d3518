import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][-1])
print(sys.argv[1], "value %.0f ms/step %.4f"%(j['value'], j['ms_per_step']), {k:round(v['total_ms']/max(v['launches'],1)*1000) for k,v in j['kernels_ms'].items()}, j.get('exact_check',{}).get('pass'))
