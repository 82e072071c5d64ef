#!/usr/bin/env python3
"""Issue-slot model of ONE wave of the trace kernel's state machine (CPU only, no GPU needed).

What it is for: pricing scheduling policies (event gate, contexts per lane) in VALU wave-instructions per ray BEFORE building them.
Costs are the static VALU counts of the bench kernel's code regions (tools/isa_regions.py over the compiler's assembly); the
ray population (steps per ray, how rays end) is synthetic, calibrated against the pass statistics of the TDT_STATS build
(profiles/r04_loss_budget.json).  A model, not a measurement: every policy it favours is then A/B-ed on the GPU."""
import argparse, random

C = dict(trav=115, hit_pro=57, lamb=258, metal=180, diel=183, hit_epi=13, end=127, fetch=130, primary=101, thr=22, newray=93, swap=14)


def sim(K=1, T=40, n_mean=10.0, p=(0.45, 0.18, 0.12, 0.25), rays=150000, seed=1, serve_parked=True, max_bounce=8, spp_rays=170, verbose=False):
    """p = (lambert, metal, dielectric, no-hit) per ray end.  K contexts per lane; context 0 of a lane is the active one."""
    rnd = random.Random(seed)
    L = 64
    # context: [steps_left, kind_waiting or None]
    def new_ray():
        # steps ~ geometric-ish with mean n_mean (min 1)
        n = 1
        q = 1.0 - 1.0 / n_mean
        while rnd.random() < q:
            n += 1
        return n
    ctx = [[[new_ray(), None, 0, rnd.randrange(spp_rays)] for _ in range(K)] for _ in range(L)]   # steps_left, waiting kind, bounce, rays until pixel end
    instr = 0; done = 0; trav_lane = 0; trav_pass = 0; ev_pass = 0; ev_lane = 0; swaps = 0
    def kind():
        r = rnd.random(); a = 0
        for i, q in enumerate(p):
            a += q
            if r < a: return i
        return 3
    while done < rays:
        # traversal pass
        active = 0; need_swap = False
        for l in range(L):
            c = ctx[l]
            if c[0][1] is not None and K > 1:
                for j in range(1, K):
                    if c[j][1] is None:
                        c[0], c[j] = c[j], c[0]; need_swap = True; break
            a = c[0]
            if a[1] is None:
                active += 1
                a[0] -= 1
                if a[0] <= 0:
                    a[1] = kind()
        instr += C['trav'] + (C['swap'] if need_swap else 0)
        swaps += 1 if need_swap else 0
        trav_pass += 1; trav_lane += active
        # gate
        if K == 1:
            waiting = sum(1 for l in range(L) if ctx[l][0][1] is not None)
        else:
            waiting = sum(1 for l in range(L) if any(c[1] is not None for c in ctx[l]))
        n_trav = sum(1 for l in range(L) if ctx[l][0][1] is None)
        if waiting < T and n_trav > 0:
            continue
        # event pass: serve one waiting context per lane
        cost = C['thr']; kinds = [0, 0, 0, 0]; served = 0; n_new = 0; n_prim = 0; n_end = 0; n_fetch = 0; sw = False
        for l in range(L):
            c = ctx[l]
            if c[0][1] is None:
                if K > 1 and serve_parked:
                    for j in range(1, K):
                        if c[j][1] is not None:
                            c[0], c[j] = c[j], c[0]; sw = True; break
                if c[0][1] is None: continue
            a = c[0]
            k = a[1]; kinds[k] += 1; served += 1
            a[1] = None; a[0] = new_ray(); done += 1
            a[3] -= 1
            if k < 3:
                a[2] += 1
                if a[2] >= max_bounce or (k == 1 and rnd.random() < 0.1): ended = True
                else: ended = False; n_new += 1
            else: ended = True
            if ended:
                n_end += 1; a[2] = 0
                if a[3] <= 0: n_fetch += 1; a[3] = spp_rays
                n_prim += 1; n_new += 1
        if sum(kinds[:3]): cost += C['hit_pro'] + C['hit_epi']
        if kinds[0]: cost += C['lamb']
        if kinds[1]: cost += C['metal']
        if kinds[2]: cost += C['diel']
        if n_end: cost += C['end']
        if n_fetch: cost += C['fetch']
        if n_prim: cost += C['primary']
        if n_new: cost += C['newray']
        if sw: cost += C['swap']
        instr += cost; ev_pass += 1; ev_lane += served
    return dict(instr_per_ray=instr / done, trav_util=trav_lane / (64.0 * trav_pass), ev_served=ev_lane / max(ev_pass, 1),
                trav_pass_per_ray=trav_pass / done, ev_pass_per_ray=ev_pass / done, swap_frac=swaps / trav_pass)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--n', type=float, default=10.0)
    ap.add_argument('--rays', type=int, default=100000)
    a = ap.parse_args()
    for K in (1, 2, 3):
        for T in (16, 24, 32, 40, 48, 56, 60, 63):
            r = sim(K=K, T=T, n_mean=a.n, rays=a.rays)
            print('K=%d T=%2d  instr/ray %6.1f  trav util %.2f  served/evpass %4.1f  swap passes %.2f' % (K, T, r['instr_per_ray'], r['trav_util'], r['ev_served'], r['swap_frac']))
