"""follow-up of tools/race25.py: clone ONE intermediate tensor per run (index on the command line list) so the schedule is
barely perturbed, and compare it with the single-stream run"""
import os, sys
sys.argv = [sys.argv[0], "0"] + sys.argv[1:]
src = open(os.path.join(os.getcwd(), "tools", "race25.py")).read().split("ref = bwd(False, False)")[0]
sys.argv = [sys.argv[0], "0"]
exec(src)
ref = bwd(False, False)
t_ref = []
bwd(False, False, t_ref)
names = [n for n, _ in t_ref]


def bwd_clone_at(idx):
    got = {}
    cnt = [0]

    def tap(name, t):
        if cnt[0] in idx:
            got[cnt[0]] = t.clone()
        cnt[0] += 1
    eng.use_side_stream, eng.block_joins, eng.debug_tap = True, False, tap
    gf = torch.empty_like(net.flat_params)
    eng.backward(sv, dlogits.clone(), gf)
    torch.cuda.synchronize()
    eng.debug_tap = None
    return gf, got


for idx in ([7],):
    for rep in range(6):
        gf, got = bwd_clone_at(idx)
        msg = []
        for i, t in got.items():
            a, b = t.float(), t_ref[i][1].float()
            nd = int((a != b).sum())
            msg.append("%s: %d elements differ (max %.3e)" % (names[i], nd, float((a - b).abs().max())))
            if nd:
                w = (a != b).nonzero()
                vox = w[:, 0].unique()
                D_, H_, W_ = 28, 64, 64
                co = [(int(v) // (D_ * H_ * W_), int(v) // (H_ * W_) % D_, int(v) // W_ % H_, int(v) % W_) for v in vox.tolist()]
                msg.append("\n   voxels (n,d,h,w): %s" % co[:80])
                v0 = int(vox[0]); chs = w[w[:, 0] == v0][:, 1].tolist()
                msg.append("\n   channels at first voxel: %s  got %s  ref %s" % (chs, a[v0, chs].tolist()[:4], b[v0, chs].tolist()[:4]))
        print("clone %s: final grads %s | %s" % (idx, "differ" if not torch.equal(gf, ref) else "equal", "; ".join(msg)))
