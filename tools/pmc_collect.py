"""Fold rocprofv3 `--pmc` counter_collection CSVs (one directory per counter pass) into one JSON: per kernel name, per counter,
launches and the average value per launch.  python tools/pmc_collect.py <out.json> <dir> [<dir> ...]"""
import csv, glob, json, os, sys
from collections import defaultdict


def fold(dirs):
    acc = defaultdict(lambda: defaultdict(lambda: defaultdict(float)))       # kernel -> counter -> dispatch id -> value
    for d in dirs:
        for f in glob.glob(os.path.join(d, '**', '*counter_collection.csv'), recursive=True):
            for r in csv.DictReader(open(f)):
                acc[r['Kernel_Name']][r['Counter_Name']][r['Dispatch_Id']] += float(r['Counter_Value'])
    out = {}
    for k, cs in acc.items():
        out[k] = {c: {'launches': len(v), 'avg': sum(v.values()) / len(v)} for c, v in cs.items()}
    return out


if __name__ == '__main__':
    json.dump(fold(sys.argv[2:]), open(sys.argv[1], 'w'), indent=1, sort_keys=True)
