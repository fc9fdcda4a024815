"""The ONE plan file (dsml_thesis_amd/igemm_plans.json: sections f32 / bf16x3 / f16x2 / ps_bf16x3 / ps_f16x2, see engine.py) and the
flat {key: [tile_cfg, splitk]} files the tuning tools write and the LDMK_*_TABLE overrides read.
   python tools/merge_plans.py --extract f16x2 gpurun_out/plans_f16x2.json      # a section as a flat file (a sweep's starting point)
   python tools/merge_plans.py --merge f16x2 gpurun_out/plans_f16x2.json        # a flat file's entries into that section
   python tools/merge_plans.py --replace f16x2 gpurun_out/plans_f16x2.json      # ... or the flat file AS the section
"""
import argparse
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PLAN_FILE = os.path.join(ROOT, "dsml_thesis_amd", "igemm_plans.json")
SECTIONS = ("f32", "bf16x3", "f16x2", "ps_bf16x3", "ps_f16x2")


def load():
    return json.load(open(PLAN_FILE))


def section(name):
    assert name in SECTIONS, name
    return dict(load().get(name, {}))


def store(name, flat, replace=False):
    assert name in SECTIONS, name
    d = load()
    if replace:
        d[name] = dict(flat)
    else:
        d.setdefault(name, {}).update(flat)
    json.dump(d, open(PLAN_FILE, "w"), indent=0, sort_keys=True)
    return len(d[name])


if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    g = ap.add_mutually_exclusive_group(required=True)
    g.add_argument("--extract", nargs=2, metavar=("SECTION", "FLAT"))
    g.add_argument("--merge", nargs=2, metavar=("SECTION", "FLAT"))
    g.add_argument("--replace", nargs=2, metavar=("SECTION", "FLAT"))
    a = ap.parse_args()
    if a.extract:
        json.dump(section(a.extract[0]), open(a.extract[1], "w"), indent=0, sort_keys=True)
        print(f"wrote {a.extract[1]}")
    else:
        name, path = a.merge or a.replace
        n = store(name, json.load(open(path)), replace=bool(a.replace))
        print(f"{PLAN_FILE}: section {name} now has {n} shapes")
