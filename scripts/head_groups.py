"""Round-5 study, review item 5: would launching the head of a step (statistics + covariance chain) PER GROUP OF CHAINS, with a ready
word per chain gating a superblock's first walk inside the ONE launch of the state chain, shorten the default-mode step?

Input: the per-chain finish times of the state chain the GPU measured in round 5 (the instrumented `k_sb_async` instances behind
CONSENRICH_AMD_SB_DEBUG printed "chains final at (us, bins)"; they were retired in round 6, the measurement is kept as
profiles/r05_state_chain_finish_times.txt; file given as argv[1], several lines are averaged).  Chains are independent once their records stand (scripts/ubench/sb_async_sim.c with per-chain ready times confirms:
a chain's finish = its ready time + the time it takes alone), so the step ends at max_c(ready[group(c)] + F_c) + tail.
head(g) = FIX + HEAD * bins(g) / bins (both head kernels are bandwidth-bound); SLOW: how much a chain's wavefronts slow down while
heads of later groups share the chip with them.  Orders: what is known before the step (length) against an oracle that knows the
previous IDENTICAL step's finish times; and, because WHICH chain is the unlucky one is a matter of rounding coincidences (DESIGN
section 11: it changes with the data, with every E-step's kappa), the expectation over that luck: each chain's excess over the
length trend is reshuffled among the chains 4000 times."""
import random
import re
import statistics
import sys

L = [1244783, 1210968, 991478, 951073, 907692, 854030, 796730, 725694, 691974, 668988, 675433, 666377, 571822, 535219, 509957,
     451692, 416288, 401887, 293088, 322221, 233550, 254093]
HEAD, FIX, TAIL, LAUNCH = 1090.0, 25.0, 790.0, 2000.0          # us: measured head, per-group launch cost, tail behind the launch, the launch


def main():
    runs = []
    for line in open(sys.argv[1]):
        if "chains final at" in line:
            runs.append([float(t.split(":")[0]) for t in re.findall(r"\d+:\d+", line)])
    F = [sum(r[c] for r in runs) / len(runs) for c in range(22)]
    k = LAUNCH / max(F)                                          # the debug instance of the kernel runs ~15 % slower than the shipped one
    F = [f * k for f in F]
    N = sum(L)
    base = HEAD + LAUNCH + TAIL
    print(f"{len(runs)} measured launches; per-chain state-chain times, scaled to a {LAUNCH:.0f}-us launch (us):")
    print("  " + " ".join(f"chr{c + 1}:{F[c]:.0f}" for c in range(22)))
    print(f"today: head {HEAD:.0f} + state chain {LAUNCH:.0f} + tail {TAIL:.0f} = {base:.0f} us per step")

    def run(order, G, slow):
        idx = sorted(range(22), key=lambda c: -order[c])
        grp, acc, g = {}, 0, 0
        for c in idx:
            grp[c] = g
            acc += L[c]
            if acc >= N * (g + 1) / G and g < G - 1:
                g += 1
        tg, t = [], 0.0
        for gg in range(G):
            t += FIX + HEAD * sum(L[c] for c in range(22) if grp[c] == gg) / N
            tg.append(t)
        end = 0.0
        for c in range(22):
            start, dur = tg[grp[c]], F[c]
            ov = min(max(tg[-1] - start, 0.0), dur * slow)
            end = max(end, start + dur + ov * (slow - 1.0) / slow)
        return end + TAIL

    for name, order in (("longest chain first (known before the step)", L),
                        ("slowest chain of the previous IDENTICAL step first (oracle)", F)):
        for G in (2, 3, 4, 6, 8):
            print(f"{name:62s} G={G}: " + "  ".join(
                f"slow x{s:.2f}: {run(order, G, s):5.0f} us ({100 * (run(order, G, s) - base) / base:+5.1f} %)" for s in (1.0, 1.15, 1.3)))

    mx, my = sum(L) / 22.0, sum(F) / 22.0
    b = sum((l - mx) * (f - my) for l, f in zip(L, F)) / sum((l - mx) ** 2 for l in L)
    a = my - b * mx
    res = [f - (a + b * l) for l, f in zip(L, F)]
    print(f"length trend of the measured times: {a:.0f} us + {b * 1e5:.1f} us per 1e5 bins; what is left (the chain's luck): sd {statistics.pstdev(res):.0f} us")
    random.seed(1)
    for G in (2, 3, 4, 6):
        gains = []
        for _ in range(4000):
            r = res[:]
            random.shuffle(r)
            Ft = [a + b * l + x for l, x in zip(L, r)]
            F[:] = Ft
            base_t = HEAD + max(Ft) + TAIL
            gains.append(100.0 * (run(L, G, 1.0) - base_t) / base_t)
        gains.sort()
        print(f"luck reshuffled, longest chain first, G={G}, no slow-down: mean {statistics.mean(gains):+.1f} %, median {gains[2000]:+.1f} %, "
              f"10th..90th percentile {gains[400]:+.1f} .. {gains[3600]:+.1f} %")


main()
