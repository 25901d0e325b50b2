import sys, re, json
arm = None
for line in open(sys.argv[1]):
    if line.startswith("=="): arm = line.strip(); clk = []
    elif line.startswith("[pair_p] block 0"):
        m = re.search(r"(\d+) shader cycles in ([0-9.]+) ms = ([0-9.]+) GHz", line); clk.append((float(m.group(2)), float(m.group(3)), int(m.group(1))))
    elif line.startswith("[pair_p diag]"): print(arm, line.strip())
    elif line.startswith("{"):
        j = json.loads(line); best = min(clk) if clk else (0, 0, 0)
        print(f"{arm:84s} phase {j['phase_ms']['filter_gemm']:.3f} ms  kernel best {best[0]:.3f} ms @ {best[1]:.3f} GHz  {best[2]} cycles -> {7.714e12 / (best[0] * 1e-3) / 2.5e15 if best[0] else 0:.3f}")
