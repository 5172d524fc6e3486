#!/usr/bin/env python3
"""Where one replay of the detect graph (batch 1) spends its time.  Two modes (tools only):
  run:      detect_graph_trace.py run [backbone] [size] [replays]   -- under rocprofv3 --kernel-trace; replays are separated by
            a marker launch (mrcnn_fill_zero of 4 096 floats on the same stream)
  report:   detect_graph_trace.py report <kernel_trace.csv> [replay]  -- kernels of one replay in start order grouped into
            phases (trunk = up to the first top-k / proposal kernel), summed kernel time, idle time between kernels"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def run():
    import torch
    from caesar_mrcnn_amd import ops
    from caesar_mrcnn_amd.config import run_py_config
    from caesar_mrcnn_amd.model import MaskRCNN
    backbone = sys.argv[2] if len(sys.argv) > 2 else "resnet101"
    size = int(sys.argv[3]) if len(sys.argv) > 3 else 256
    n = int(sys.argv[4]) if len(sys.argv) > 4 else 6
    cfg = run_py_config(backbone=backbone, imgsize=size, mode="inference")
    m = MaskRCNN("inference", cfg, "/tmp/x", device=torch.device("cuda:0"))
    x = torch.rand(1, size, size, 3, device="cuda") * 255
    w = torch.tensor([[0., 0., 1., 1.]], device="cuda")
    marker = torch.empty(7, dtype=torch.uint8, device="cuda")    # 7 bytes: fill_zero_bytes_kernel, nothing else launches it
    for _ in range(3):
        m.engine.infer_graphed(x, w)
    torch.cuda.synchronize()
    for _ in range(n):
        ops.fill_zero(marker)
        m.engine.infer_graphed(x, w)
    ops.fill_zero(marker)
    torch.cuda.synchronize()


def report():
    import csv
    rows = list(csv.DictReader(open(sys.argv[2])))
    k = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    ev = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].split('(')[0].replace('void ', ''),
                 int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z']) // max(1, int(r['Workgroup_Size_X']))) for r in rows)
    seps = [i for i, e in enumerate(ev) if e[2].startswith('fill_zero_bytes')]
    a, b = seps[-(k + 2)], seps[-(k + 1)]
    seg = ev[a + 1:b]
    t0, t1 = seg[0][0], seg[-1][1]
    print("replay: %d launches, first start .. last end %.3f ms, marker to marker %.3f ms" % (len(seg), (t1 - t0) / 1e6, (ev[b][0] - ev[a][0]) / 1e6))
    names = [e[2] for e in seg]

    def first(pred):
        for i, n in enumerate(names):
            if pred(n):
                return i
        return len(names)
    i_prop = first(lambda n: 'topk' in n or 'select_sort' in n or 'proposal' in n or 'rpn_softmax' in n or 'softmax' in n)
    i_roi = first(lambda n: n.startswith('roialign'))
    i_det = first(lambda n: 'detection' in n)
    cuts = [("trunk + RPN convolutions", 0, i_prop), ("ProposalLayer", i_prop, i_roi), ("class head", i_roi, i_det),
            ("DetectionLayer + mask head", i_det, len(seg))]
    for name, lo, hi in cuts:
        if hi <= lo:
            continue
        part = seg[lo:hi]
        busy = sum(e[1] - e[0] for e in part)
        wall = part[-1][1] - part[0][0]
        print("  %-28s %4d launches  wall %.3f ms  kernel time %.3f ms  idle %.3f ms  (%.1f us per launch)" % (
            name, len(part), wall / 1e6, busy / 1e6, (wall - busy) / 1e6, wall / 1e3 / len(part)))
    for name, lo, hi in (cuts if len(sys.argv) > 4 and sys.argv[4] == "all" else cuts[1:]):      # report <trace> <replay> all: the trunk's launches too
        print("  %s kernels in start order:" % name)
        for s_, e_, n_, wg_ in seg[lo:hi]:
            print("    +%8.1f us  %7.1f us  wgs=%-6d %s" % ((s_ - t0) / 1e3, (e_ - s_) / 1e3, wg_, n_[:60]))
    import collections
    acc, cnt = collections.Counter(), collections.Counter()
    for s, e, n, wg in seg[:i_prop]:
        acc[(n[:44], wg)] += e - s
        cnt[(n[:44], wg)] += 1
    print("  trunk kernels:")
    for (n, wg), t in acc.most_common(16):
        print("    %-44s wgs=%-6d n=%3d  sum %.3f ms  mean %.1f us" % (n, wg, cnt[(n, wg)], t / 1e6, t / 1e3 / cnt[(n, wg)]))
    gaps = sorted(((seg[i + 1][0] - seg[i][1]) / 1e3 for i in range(min(i_prop, len(seg) - 1))))
    if gaps:
        print("  trunk gaps between consecutive launches: median %.2f us, p90 %.2f us, max %.1f us, sum %.3f ms" % (
            gaps[len(gaps) // 2], gaps[int(len(gaps) * 0.9)], gaps[-1], sum(g for g in gaps if g > 0) / 1e3))


if __name__ == "__main__":
    (run if sys.argv[1] == "run" else report)()
