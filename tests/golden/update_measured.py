#!/usr/bin/env python
"""Collects the 'MEASURED <case> <arithmetic> <max-norm vs o64> <max-norm vs o32>' lines that tests/test_configs_gpu.py prints for its
deterministic-mode steps (run with `pytest -s` on the GPU box) into tests/golden/measured_max_norm.json -- the values the imgs_pred
max-norm gates are derived from (measured + 30 %).  Data only; the deterministic mode is reproducible across boxes.

    python tests/golden/update_measured.py gpurun_out/gpu_tests.log [more logs ...]
"""
import json
import os
import re
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, 'measured_max_norm.json')


def main():
    # (start from scratch: the table describes ONE build -- pass every log of that build)
    res = {}
    n = 0
    for path in sys.argv[1:]:
        for line in open(path, errors='replace'):
            m = re.search(r'MEASURED (\S+) (\S+) ([0-9.eE+-]+) ([0-9.eE+-]+)', line)
            if m:
                case, prec, a, b = m.group(1), m.group(2), float(m.group(3)), float(m.group(4))
                if case.endswith('bp'):          # chaotic feedback case: floor-scaled gates only (tests/test_configs_gpu.py)
                    continue
                cur = res.setdefault(case, {}).get(prec)
                # several logs: keep the largest value seen (the bound must admit every run of the same build)
                res[case][prec] = {'o64': max(a, cur['o64']) if cur else a, 'o32': max(b, cur['o32']) if cur else b}
                n += 1
    json.dump(res, open(OUT, 'w'), indent=1, sort_keys=True)
    print('%d lines -> %s' % (n, OUT))


if __name__ == '__main__':
    main()
