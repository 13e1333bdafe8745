#!/usr/bin/env python3
"""Turn a GPU parity report (gpurun_out/parity_report_gpu.json, written by tests/conftest.py) into tests/golden/parity_pins.json.

Every float comparison of the GPU suite goes through tests/helpers.py::close(), which records what was ACHIEVED next to what the
test wrote as its tolerance.  Wherever the written tolerance is more than 10x what was achieved on MI355X, the achieved figure is
pinned here and close() additionally asserts  max |a - b| <= 4 x pinned  on GPU runs -- so no comparison of the suite is held to
less than 4x what the hardware actually delivers, without hand-editing a hundred call sites after every kernel change.
Comparisons whose outcome depends on fp32 near-ties inside the victim (PCT's max-pool winners, tests/test_gpu_configs.py) keep
their written bounds: those are already set from measurement and vary with the host's BLAS threading.

    python tools/make_parity_pins.py gpurun_out/parity_report_gpu.json [more reports ...] > tests/golden/parity_pins.json

Several reports (different boxes / runs) are merged by taking the LARGEST achieved error per comparison;
``--keep tests/golden/parity_pins.json`` also merges the pins already committed the same way (box-to-box variation of earlier
runs stays covered; a comparison that got worse than 4x its committed pin is listed on stderr first)."""
import json
import sys

SLACK = 10.0
SKIP = ('cfg5', 'raw relative L2', 'fraction of elements')
# Second rule (round 5): a comparison the first rule leaves alone (its relative bound looked tight because one
# element of the expectation is near zero) but whose EFFECTIVE tolerance -- atol + rtol x |expected|, as assert_allclose applies it
# -- is more than LOOSE x what was achieved gets a pin of max(achieved, effective tolerance / FLOOR): close() then holds it to
# 4 x that, i.e. to the larger of 4 x achieved and 1/50 of what the test wrote.  (A floor, because a pin of one ulp taken from
# one box would turn the next box's two ulps into a red test.)
LOOSE, FLOOR = 50.0, 200.0


def main(paths):
    pins, kept = {}, {}
    if paths and paths[0] == '--keep':
        with open(paths[1]) as f:
            kept = json.load(f)
        paths = paths[2:]
    for path in paths:
        with open(path) as f:
            report = json.load(f)
        for test, v in report.items():
            rows = v['rows']
            if v['comparisons'] != len(rows):
                # the report keeps only the 12 worst rows of a long test: positional names (cmpN) have lost their order, NAMED
                # comparisons can still be pinned
                rows = [r for r in rows if not (r['what'].startswith('cmp') and r['what'][3:].isdigit())]
            for i, r in enumerate(rows):
                what = r['what']
                if any(s in what for s in SKIP) or r['max_abs'] <= 0.0:
                    continue
                if r['rtol'] == 0:
                    ratio = r['atol'] / r['max_abs']
                else:
                    ra = r['rtol'] / r['max_rel'] if r['max_rel'] > 0 else float('inf')
                    ratio = min(ra, r['atol'] / r['max_abs'] if r['atol'] > 0 else float('inf'))
                    if ratio == float('inf'):
                        ratio = r['rtol'] * 1.0 / max(r['max_abs_over_scale'], 1e-300)
                if ratio <= SLACK:
                    # the tolerance at the element that decides: atol + rtol |b|, with |b| ~ max_abs / max_rel there
                    eff = r['atol'] + (r['rtol'] * r['max_abs'] / r['max_rel'] if r['max_rel'] > 0 else 0.0)
                    if r.get('statistic_only') or eff <= LOOSE * r['max_abs']:
                        continue
                    slot = pins.setdefault(test, {}).setdefault(what, dict(max_abs=0.0, written_rtol=r['rtol'], written_atol=r['atol'],
                                                                           floored=True))
                    slot['max_abs'] = max(slot['max_abs'], r['max_abs'], eff / FLOOR)
                    continue
                slot = pins.setdefault(test, {}).setdefault(what, dict(max_abs=0.0, written_rtol=r['rtol'], written_atol=r['atol']))
                slot['max_abs'] = max(slot['max_abs'], r['max_abs'])
    for test, slots in kept.items():
        for what, slot in slots.items():
            mine = pins.setdefault(test, {}).get(what)
            if mine is not None and mine['max_abs'] > 4 * slot['max_abs']:
                sys.stderr.write("worse than 4x the committed pin: %s / %s: %.3g -> %.3g\n" % (test, what, slot['max_abs'], mine['max_abs']))
            if mine is None or mine['max_abs'] < slot['max_abs']:
                pins[test][what] = slot
    json.dump(pins, sys.stdout, indent=1, sort_keys=True)
    sys.stdout.write("\n")


if __name__ == '__main__':
    main(sys.argv[1:])
