"""Per run: frame rate; the LAST timed frame's blend workgroups (entry offsets after the first one's, durations) and edge-kernel
workgroups (first entry, last exit) on the device's wall clock, relative to the blend's first entry; the stage stamps of the last call."""
import sys
import numpy as np
for p in sys.argv[1:]:
    z = np.load(p)
    wg, ring, fps = z["wg"].astype(np.int64), z["ring"].astype(np.int64), float(z["fps"])
    b = wg[1]; b = b[b[:, 6] > 0]
    e = wg[2][:8000]; e = e[e[:, 0] > 0]
    g = wg[2][8190]; ib = wg[2][8189][0]
    t0 = b[:, 6].min()
    us = lambda x: (x - t0) / 100.0
    ent, ex = us(b[:, 6]), us(b[:, 7])
    dur = ex - ent
    last = ring[np.argmax(ring[:, 0])]
    print("%s fps %.0f | blend: %d wgs, entry offsets p50 %.1f p90 %.1f max %.1f us; duration p50 %.1f p90 %.1f max %.1f; last exit %.1f | edge kernel OF THE PREVIOUS CALL entries %.1f .. exits %.1f (last), dense(>768 entries) exits p50 %.1f"
          % (p.split("_")[-1][:-4], fps, len(b), np.percentile(ent, 50), np.percentile(ent, 90), ent.max(), np.percentile(dur, 50), np.percentile(dur, 90), dur.max(), ex.max(),
             us(e[:, 0].min()), us(e[:, 1].max()), np.percentile(us(e[e[:, 3] > 768, 1]), 50) if (e[:, 3] > 768).any() else -1))
    print("      gate: entry %.1f exit %.1f (polls %d) | integrate's first workgroup %.1f  => last blend exit -> gate exit %.1f us, gate exit -> integrate %.1f us"
          % (us(g[0]), us(g[1]), g[2], us(ib), us(g[1]) - ex.max(), us(ib) - us(g[1])))
