import json,sys
j=json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][-1])
print(j["value"], j["ms_per_step"], j["roofline"]["frac"], j["qerror_check"]["pass"], j["qerror_check"]["rel_delta"], j["full_run"]["seconds"])
