"""Static instruction budget of k_detect: every instruction of one kernel instance attributed to a PHASE of the source
(through the .loc line tables of a -gline-tables-only -save-temps build) and to an instruction CLASS.

    hipcc ... -gline-tables-only -save-temps gr_adsb_amd/csrc/adsb_hip.hip   (tools/r06_budget.sh does it)
    python tools/isa_budget.py <gfx950 .s file> '_ZN4adsb8k_detectILi5ELi1EEEvNS_10DetectArgsE' [more symbols]

Static counts are code size per phase, not executed instructions (loops, branches): the executed totals per tile come from
the SQ_INSTS_* counters of the shipped kernel and of phase-ablated side copies (profiles/r06_instruction_budget.txt holds
both).  A leaf location inside the HIP headers (__ballot, __popcll, __shfl ...) inherits the phase of the instruction in
front of it."""
import re
import sys
from collections import OrderedDict, defaultdict


def phases_of(device_h):
    """(lo, hi, phase) line ranges of gr_adsb_amd/csrc/adsb_device.h, found by the functions' own text."""
    src = open(device_h).read().splitlines()

    def line_of(pat, start=0):
        for i in range(start, len(src)):
            if pat in src[i]:
                return i + 1
        raise KeyError(pat)

    marks = [
        ("mag2f(float re", "convert"), ("struct BurstFetch", "long pulses / global gathers"),
        ("float noise_median(", "record: median"), ("void rec_store_head(", "record: stores / stage"),
        ("void burst_reduce(", "long pulses / global gathers"), ("void burst_from_window(", "record: window, peak, control"),
        ("constexpr int kStage = 16;", "record: stores / stage"), ("void pend_step(", "record: pending bursts"),
        ("void pend_flush(", "record: pending bursts"), ("struct Body {", "loads (issue)"),
        ("void body_convert(", "convert"), ("int body_commit(", "commit: LDS stores, max, reload"),
        ("unsigned unit_mask(", "mask units"), ("bool chips_match(", "per rise: fall, centre, chips"),
        ("void detect_body(", "setup / epilogue"), ("auto head_fill =", "setup / epilogue"),
        ("auto rises_to_records =", "per rise: fall, centre, chips"), ("// -- C: this tile's hits", "hit loop: list word, control"),
        ("auto process_tile =", "mask units"), ("// -- P: bursts met in earlier tiles", "record: pending bursts"),
        ("// -- B: rises among the tile's own samples", "rise list (mask algebra, scan)"),
        ("if (n_pend > 0 && it + 1 == ntile)", "record: pending bursts"),
        ("// what the next tile inherits", "slide"), ("const int lane_outer = lane;", "tile loop control"),
        ("// what the stage still holds", "setup / epilogue"), ("void longrun_entry(", "other kernels"),
    ]
    # slice_window sits between the PendList structs and burst_from_window
    marks.append(("void slice_window(", "record: bit slices"))
    pts = sorted((line_of(p), ph) for p, ph in marks)
    out = []
    for (lo, ph), (hi, _) in zip(pts, pts[1:] + [(len(src) + 1, None)]):
        out.append((lo, hi, ph))
    return out


HIP_HELPERS = {  # gr_adsb_amd/csrc/adsb_hip.hip leaf lines (the inline-asm helpers) -> phase
    "adsb_above4": "mask units", "adsb_mag2": "convert", "adsb_sdot4": "convert", "adsb_fmax3": "per rise: fall, centre, chips",
    "adsb_wave_incl_scan": "rise list (mask algebra, scan)", "adsb_lane_up1": "rise list (mask algebra, scan)",
    "adsb_wave_min_u32": "record: median", "adsb_wave_max_u32": "record: median", "adsb_ld_stream": "loads (issue)",
    "adsb_st_stream": "record: stores / stage", "adsb_cold": None, "adsb_opaque": None, "adsb_after": "commit: LDS stores, max, reload",
    "adsb_setprio": "rise list (mask algebra, scan)",
}


def hip_ranges(hip_path):
    src = open(hip_path).read().splitlines()
    out = []
    for name, ph in HIP_HELPERS.items():
        for i, l in enumerate(src):
            if re.search(r"\b%s\s*\(" % name, l) and ("__device__" in l or "__device__" in src[max(0, i - 1)]):
                j = i
                while j < len(src) and not src[j].startswith("}"):
                    j += 1
                out.append((i + 1, j + 2, ph))
                break
    return out


def klass(m):
    if m.startswith("ds_"):
        return "LDS"
    if m.startswith(("global_", "buffer_", "flat_", "scratch_")):
        return "VMEM"
    if m.startswith("s_load") or m.startswith("s_buffer_load"):
        return "SMEM"
    if m.startswith(("s_cbranch", "s_branch", "s_setpc", "s_swappc", "s_endpgm")):
        return "BRANCH"
    if m.startswith("s_waitcnt"):
        return "wait"
    if m in ("s_nop", "s_sleep", "s_setprio", "s_barrier"):
        return "nop/prio"
    if m.startswith("s_"):
        return "SALU"
    if m.startswith("v_"):
        return "VALU"
    return "other"


def main():
    asm, syms = sys.argv[1], sys.argv[2:]
    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    dev_ranges = phases_of(root + "/gr_adsb_amd/csrc/adsb_device.h")
    hip_rng = hip_ranges(root + "/gr_adsb_amd/csrc/adsb_hip.hip")
    lines = open(asm).read().splitlines()
    files = {}
    for l in lines:
        m = re.match(r'\s*\.file\s+(\d+)\s+"([^"]*)"\s+"([^"]*)"', l)
        if m:
            files[int(m.group(1))] = m.group(3)
    classes = ["VALU", "SALU", "LDS", "VMEM", "SMEM", "BRANCH", "wait", "nop/prio"]
    for sym in syms:
        start = next(i for i, l in enumerate(lines) if l.startswith(sym + ":"))
        tab = OrderedDict()
        cur = "setup / epilogue"
        n = 0
        for l in lines[start + 1:]:
            t = l.strip()
            if t.startswith(".Lfunc_end") or t.startswith(".end_amdhsa_kernel"):
                break
            m = re.match(r"\.loc\s+(\d+)\s+(\d+)", t)
            if m:
                f, ln = files.get(int(m.group(1)), ""), int(m.group(2))
                ph = False
                if f == "adsb_device.h":
                    ph = next((p for lo, hi, p in dev_ranges if lo <= ln < hi), False)
                elif f == "adsb_hip.hip":
                    ph = next((p for lo, hi, p in hip_rng if lo <= ln < hi), False)
                if ph:                      # None / False: inherit
                    cur = ph
                continue
            if not t or t.startswith((";", ".", "//")) or t.endswith(":"):
                continue
            mn = t.split()[0]
            if not re.match(r"^[a-z_0-9]+$", mn):
                continue
            tab.setdefault(cur, defaultdict(int))[klass(mn)] += 1
            n += 1
        print("## %s: %d instructions (static)" % (sym, n))
        print("%-36s " % "phase" + " ".join("%7s" % c for c in classes) + "   total")
        tot = defaultdict(int)
        for ph, row in sorted(tab.items(), key=lambda kv: -sum(kv[1].values())):
            print("%-36s " % ph + " ".join("%7d" % row.get(c, 0) for c in classes) + "  %6d" % sum(row.values()))
            for c in classes:
                tot[c] += row.get(c, 0)
        print("%-36s " % "all" + " ".join("%7d" % tot[c] for c in classes) + "  %6d" % sum(tot.values()))
        print()


if __name__ == "__main__":
    main()
