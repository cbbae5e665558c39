"""Summary of the TE_MSM_TRACE_HOST stamps of tools/trace_bound_tickets.py (the part after "==== timed loop"): per ticket piece, the time
of the upload call, of the wait behind it and of enqueueing the kernels (sorted, us).  python tools/stamp_summary.py stamps.txt ..."""
import re, sys
for f in sys.argv[1:]:
    t = {}; up = []; aw = []; enq = []; on = False
    for l in open(f):
        if l.startswith("==== timed loop"):
            on = True
        m = re.match(r'\[scalar slice ws (\d+)\]\s+([\d.]+) us  (upload begins|upload call returned|upload awaited|piece enqueued) (\d+)', l)
        if not (on and m):
            continue
        ws, tt, what, i = int(m.group(1)), float(m.group(2)), m.group(3), int(m.group(4))
        if what == 'upload begins':
            t[(ws, i)] = tt
        elif (ws, i) in t:
            (up if what == 'upload call returned' else aw if what == 'upload awaited' else enq).append(tt - t[(ws, i)]); t[(ws, i)] = tt
    print(f)
    for nm, a in (('upload call', up), ('wait behind it', aw), ('kernels enqueued', enq)):
        a.sort()
        print("  %-17s n=%d  median %d  mean %d  max %d   deciles %s" % (nm, len(a), a[len(a) // 2] if a else 0, sum(a) / max(1, len(a)), a[-1] if a else 0,
                                                                        ' '.join('%d' % a[min(len(a) - 1, len(a) * k // 10)] for k in range(1, 10)) if a else ''))
