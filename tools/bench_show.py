"""print the parts of a bench.py line that the round's targets are stated on: tools/bench_show.py <file.json>"""
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print("value %.4g %s, ms_per_step %.2f, roofline %s" % (d["value"], d["unit"], d["ms_per_step"], {k: d["roofline"][k] for k in ("frac", "achieved", "traffic") if k in d["roofline"]}))
k = d.get("kmer_prefilter")
if k:
    print("kmer:", {x: k.get(x) for x in ["queries_per_s", "ms_per_query", "prefilter_device_ms_per_query", "prefilter_device_ms_per_query_solo", "host_threads"]})
    print("  stages:", {a: round(b, 3) for a, b in k["stage_ms_per_batch32_solo"].items()})
    print("  partition:", k.get("segments_solo"))
    r = k["roofline"]; print("  roofline:", {x: r.get(x) for x in ("frac", "achieved", "traffic", "kernel_ms")}, "solo", r.get("solo"))
    print("  align:", k.get("align_roofline"))
a = d.get("allvsall")
if a:
    print("allvsall:", {x: a.get(x) for x in ("queries_per_s", "host_wall_ms_per_batch")})
    print("  module:", a.get("native_module_end_to_end"))
    print("  roofline:", a.get("roofline"))
t = d.get("align_type2")
if t: print("type2:", {x: t.get(x) for x in ("ms_per_query", "value")}, t.get("align_roofline"))
s = d.get("single_query_100k")
if s: print("single:", {x: s.get(x) for x in s if "ms" in x})
print("align_roofline:", d.get("align_roofline"))
print("phase_wall_s:", d.get("phase_wall_s"))
