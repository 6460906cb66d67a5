"""Python restatement of plan_dw (csrc/effnet.hip) to look at the tile plans; usage: python tools/exp/dw_plan.py"""
def plan(C, OH, OW, K, S, esize=2, xp=False, budget=None, stage_cost=None):
    budget = budget or 48 * 1024
    stage_cost = stage_cost or (96.0 if xp else 6.0)
    V = 16 // esize
    chunks = C // V
    LPP = next(d for d in range(8, 0, -1) if chunks % d == 0)
    CS = LPP * V
    OXT = 4 if OW % 4 == 0 else 3 if OW % 3 == 0 else 5 if OW % 5 == 0 else 4
    nxg = -(-OW // OXT)
    pitch16 = LPP | 1
    fixed = (K * K + 2) * CS * 4
    lanes = 256 // LPP
    ib_f = lambda th, twg: ((th - 1) * S + K) * ((twg * OXT - 1) * S + K) * pitch16 * 16
    best = None
    for th in range(1, OH + 1):
        ty = -(-OH // th)
        if -(-OH // ty) != th: continue
        for twg in range(1, nxg + 1):
            tx = -(-nxg // twg)
            if -(-nxg // tx) != twg: continue
            ib = ib_f(th, twg)
            if ib + fixed > budget: break
            groups = th * twg
            pg = min(groups, lanes)
            imb = 256 // (LPP * pg)
            while imb > 1 and imb * ib + fixed > budget: imb -= 1
            imb = max(imb, 1)
            passes = -(-groups // pg)
            staged = imb * ((th - 1) * S + K) * ((twg * OXT - 1) * S + K) * LPP
            cost = (staged * stage_cost + passes * 256 * K * K * OXT + 600.0) / (imb * groups * OXT * LPP)
            if best is None or cost < best[0]:
                best = (cost, dict(TH=th, TWG=twg, IMB=imb, PG=pg, tiles=ty * tx, LPP=LPP, CS=CS, OXT=OXT, passes=passes,
                                   lds=imb * ib + fixed, win=((th - 1) * S + K, (twg * OXT - 1) * S + K),
                                   halo=((th - 1) * S + K) * ((twg * OXT - 1) * S + K) / (th * twg * OXT * S * S)))
    return best
if __name__ == "__main__":
    for name, (C, O, K, S) in {"b2": (144, 36, 3, 2), "b3": (192, 36, 3, 1), "b5": (192, 18, 5, 2), "b6": (288, 18, 5, 1), "b8": (288, 9, 3, 2)}.items():
        print(name, "loaded", plan(C, O, O, K, S))
        print(name, "fused ", plan(C, O, O, K, S, xp=True))
