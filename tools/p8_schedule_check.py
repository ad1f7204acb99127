"""Hazard check of gemm_bf16_p8_kernel's schedule (csrc/gemm.hip) on a barrier-epoch model — a dev tool that runs on the CPU.

Model.  Every wave executes the kernel's event list in order; `BAR` advances the wave's epoch; all waves take part in every barrier, so an
event a of wave V happens-before an event b of wave W (V != W) iff epoch(a) < epoch(b); inside a wave, program order.
  * ISSUE(slot, ver): the wave starts its two LDS-DMA loads of half-tile version `ver` into `slot`.  They are known complete only after a
    later WAITVM(n) of the SAME wave for which at least... the loads are older than the wave's n most recent loads.
  * READ(slot, ver): ds_reads of that slot; known complete only after the wave's next LGKM0.
RAW: at READ(slot, ver) by W, for every wave V the completing WAITVM of V's ISSUE(slot, ver) must happen-before the READ.
WAR: at ISSUE(slot, ver+2) by V (same slot), for every wave W the LGKM0 completing W's READs of (slot, ver) must happen-before the ISSUE.
The event lists below mirror the kernel source line by line (prologue, stagger barrier, four phases per K-tile, tail conditions)."""
import sys

KINDS = ("A0", "A1", "B0", "B1")


def program(late: bool, nk: int):
    ev = []
    for st in range(2):                                  # prologue: stages 0 and 1 in steady-state order; only stage 0 is waited for
        if st < nk:
            for kind in ("A0", "B0", "B1", "A1"):
                ev.append(("ISSUE", (st, kind), st))
    ev += [("WAITVM", 8 if nk > 1 else 0), ("LGKM0",), ("BAR",)]
    if late:
        ev.append(("BAR",))
    for kt in range(nk):
        st, steady, next1 = kt & 1, kt + 2 < nk, kt >= 1 and kt + 1 < nk
        vm = 6 if steady else 0

        def seg():
            return [("BAR",), ("LGKM0",), ("MMA",), ("BAR",)]
        ev += [("READ", (st, "B0"), kt), ("READ", (st, "A0"), kt)]                       # P0
        if next1:
            ev.append(("ISSUE", (st ^ 1, "B1"), kt + 1))
        ev.append(("WAITVM", vm)); ev += seg()
        ev.append(("READ", (st, "B1"), kt))                                             # P1
        if next1:
            ev.append(("ISSUE", (st ^ 1, "A1"), kt + 1))
        ev += seg()
        ev.append(("READ", (st, "A1"), kt))                                             # P2
        if steady:
            ev.append(("ISSUE", (st, "A0"), kt + 2))
        ev.append(("WAITVM", vm)); ev += seg()
        if steady:                                                                      # P3
            ev.append(("ISSUE", (st, "B0"), kt + 2))
        ev.append(("WAITVM", vm)); ev += seg()
    if not late:
        ev.append(("BAR",))
    return ev


def annotate(ev):
    """-> per event: epoch; for ISSUE the (epoch, index) of the WAITVM that guarantees completion; for READ that of the next LGKM0"""
    epoch, out, issues, reads = 0, [], [], []
    for i, e in enumerate(ev):
        rec = {"ev": e, "epoch": epoch, "idx": i, "done": None}
        if e[0] == "BAR":
            epoch += 1
        elif e[0] == "ISSUE":
            issues.append(rec)
        elif e[0] == "READ":
            reads.append(rec)
        elif e[0] == "WAITVM":
            keep = e[1] // 2                      # half-tiles (2 loads each) allowed to stay in flight
            for r in (issues[:len(issues) - keep] if keep else issues):
                if r["done"] is None:
                    r["done"] = (epoch, i)
        elif e[0] == "LGKM0":
            for r in reads:
                if r["done"] is None:
                    r["done"] = (epoch, i)
        out.append(rec)
    return out


def before(a_epoch, a_idx, a_wave, b_epoch, b_idx, b_wave):
    return a_idx < b_idx if a_wave == b_wave else a_epoch < b_epoch


def check(nk: int):
    waves = {w: annotate(program(w >= 4, nk)) for w in range(8)}
    nbar = {w: sum(1 for r in waves[w] if r["ev"][0] == "BAR") for w in waves}
    assert len(set(nbar.values())) == 1, f"barrier counts differ: {nbar}"
    errs = 0
    for w, evs in waves.items():
        for r in evs:
            kind = r["ev"][0]
            if kind == "READ":
                slot, ver = r["ev"][1], r["ev"][2]
                for v, evs_v in waves.items():
                    iss = [x for x in evs_v if x["ev"][0] == "ISSUE" and x["ev"][1] == slot and x["ev"][2] == ver]
                    assert len(iss) == 1, (nk, slot, ver, "issued", len(iss), "times by wave", v)
                    d = iss[0]["done"]
                    if d is None or not before(d[0], d[1], v, r["epoch"], r["idx"], w):
                        errs += 1; print(f"nk={nk} RAW: wave {w} reads {slot} v{ver} (epoch {r['epoch']}) before wave {v}'s loads are known complete ({d})")
            elif kind == "ISSUE" and r["ev"][2] >= 2:
                slot, ver = r["ev"][1], r["ev"][2]
                for v, evs_v in waves.items():
                    for x in evs_v:
                        if x["ev"][0] == "READ" and x["ev"][1] == slot and x["ev"][2] == ver - 2:
                            d = x["done"]
                            if d is None or not before(d[0], d[1], v, r["epoch"], r["idx"], w):
                                errs += 1; print(f"nk={nk} WAR: wave {w} restages {slot} v{ver} (epoch {r['epoch']}) before wave {v} finished reading v{ver - 2} ({d})")
    return errs


def check_stream(nk: int, my_tiles: int) -> int:
    """gemm_bf16_p8p_kernel's bookkeeping: with the per-kind (K-tile, tile) counters of P9_ISSUE, does every slot hold the half-tile of
    the (tile, K-tile) that is being consumed when it is read?  Returns the number of mismatches."""
    gtot = my_tiles * nk
    ktc = {k: 0 for k in KINDS}
    tjc = {k: 0 for k in KINDS}
    slot, bad = {}, 0

    def issue(st, kind):
        slot[(st, kind)] = (tjc[kind], ktc[kind]) if tjc[kind] < my_tiles else None
        ktc[kind] += 1
        if ktc[kind] == nk:
            ktc[kind] = 0
            tjc[kind] += 1
    for v in range(2):
        if v < gtot:
            for kind in ("A0", "B0", "B1", "A1"):
                issue(v, kind)
    for g in range(gtot):
        st, steady, next1 = g & 1, g + 2 < gtot, g >= 1 and g + 1 < gtot
        want = (g // nk, g % nk)
        bad += slot[(st, "A0")] != want or slot[(st, "B0")] != want       # P0 reads
        if next1:
            issue(st ^ 1, "B1")
        bad += slot[(st, "B1")] != want                                    # P1 read
        if next1:
            issue(st ^ 1, "A1")
        bad += slot[(st, "A1")] != want                                    # P2 read
        if steady:
            issue(st, "A0")
            issue(st, "B0")                                                # (P3)
    return bad


if __name__ == "__main__":
    total = sum(check(nk) for nk in range(1, 12))
    total += sum(check_stream(nk, mt) for nk in range(1, 9) for mt in range(1, 6))
    print("hazards + stream mismatches:", total)
    sys.exit(1 if total else 0)
