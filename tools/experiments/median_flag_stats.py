# CPU emulation of k_median_cols' window passes on the photograph chain: per link, per tile: distinct values in the footprint, the first window's base, pixels flagged below / above
import sys, time, numpy as np
sys.path.insert(0, '/root/repo'); sys.path.insert(0, '/root/repo/tests')
import oracle_lib as O
from poppy_amd import synth
W, H = 1920, 1080
kind = sys.argv[1] if len(sys.argv) > 1 else 'photo'
img = synth.photo_pair(W, H)[0] if kind == 'photo' else synth.textured_bgr(W, H, 5)
cur = np.ascontiguousarray(img[:, :, 1])     # (the chain runs on the grey image; green stands in)
Cw, rows = 128, 22
tx, ty = (W + Cw - 1) // Cw, (H + rows - 1) // rows
for i in range(1, 12):
    k = 8 * i + 1; r = k // 2
    t0 = time.time(); out = O.median_blur_u8(cur, k)
    heavy = 0; need2 = 0; need3 = 0; flagged_hist = []
    for by in range(ty):
        for bx in range(tx):
            X0, Y0 = bx * Cw, by * rows
            fx0, fx1 = max(X0 - r, 0), min(X0 + Cw - 1 + r, W - 1)
            fy0, fy1 = max(Y0 - r, 0), min(Y0 + rows - 1 + r, H - 1)
            # the kernel unions the presence maps of the SOURCE TILES the footprint touches (coarser than the footprint itself)
            sx0, sx1 = fx0 // Cw * Cw, min((fx1 // Cw + 1) * Cw, W)
            sy0, sy1 = fy0 // rows * rows, min((fy1 // rows + 1) * rows, H)
            vals = np.unique(cur[sy0:sy1, sx0:sx1])
            D = len(vals)
            if D <= 129: continue
            heavy += 1
            tile_in = cur[Y0:min(Y0 + rows, H), X0:min(X0 + Cw, W)]
            tile_out = out[Y0:min(Y0 + rows, H), X0:min(X0 + Cw, W)]
            rank = np.searchsorted(vals, tile_out)
            # sample: 512 pixels of the tile -> their median rank - 64, clamped to [0, D - 129]
            srank = np.searchsorted(vals, tile_in[::max(1, tile_in.shape[0] // 4), :].ravel()[:512])
            base = int(np.clip(np.median(srank) - 64, 0, D - 129))
            below = int(np.count_nonzero(rank <= base)) if base > 0 else 0
            above = int(np.count_nonzero(rank >= base + 128)) if base < D - 129 else 0
            if below or above: need2 += 1
            if below and above: need3 += 1
            if below or above: flagged_hist.append(below + above)
    fh = np.array(flagged_hist) if flagged_hist else np.array([0])
    print(f"ksize {k}: tiles {tx*ty}, heavy (>129 values) {heavy}, needing a 2nd pass {need2}, a 3rd {need3}; flagged pixels per such tile: median {int(np.median(fh))}, max {int(fh.max())}, tiles with <= 32 flagged: {int((fh <= 32).sum())} ({time.time()-t0:.0f} s)", flush=True)
    cur = out
