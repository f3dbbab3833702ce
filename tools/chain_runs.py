import sys, collections
for f in sys.argv[1:]:
    probs = {}
    runs = collections.defaultdict(list)
    for line in open(f):
        t = line.split()
        if t[0] == 'P':
            probs[int(t[1])] = (t[2], int(t[-1]))
        elif t[0] == 'T':
            wg, idx, prob, mi = int(t[1]), int(t[2]), int(t[3]), int(t[4])
            name, fused = probs[prob]
            n = int(t[12]) if len(t) > 12 else 0
            if fused == 3 and 'LocalLayer_' in name and n > 0:
                fetched, ready, fin = float(t[9]), float(t[10]), float(t[11])
                runs[(mi, n)].append((fin - ready) / n)
    print(f)
    for k in sorted(runs):
        v = sorted(runs[k])
        print("  mi %d run length %d: %3d runs, per-tile median %.1f min %.1f max %.1f us" % (k[0], k[1], len(v), v[len(v)//2], v[0], v[-1]))
