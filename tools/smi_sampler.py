"""Sample GPU clocks / power with rocm-smi while a command runs; print the command's output untouched and write one summary
line (median / min / max of sclk, mclk, socket power) to the file given.   python tools/smi_sampler.py OUT -- cmd ..."""
import json
import re
import statistics
import subprocess
import sys
import threading
import time


def _sysfs_paths():
    """(sclk, mclk, power) sysfs files of the GPU(s) this container was given (sysfs shows every GPU of the host, and which
    card is ours differs from box to box); if that cannot be told, of all of them -- main() then keeps the card whose
    shader clock went highest during the run."""
    import glob
    import os
    out = []
    # the GPU(s) this container was given: the render nodes under /dev/dri, mapped to their PCI devices
    mine = set()
    for node in glob.glob("/dev/dri/renderD*"):
        d = "/sys/class/drm/%s/device" % os.path.basename(node)
        if os.path.exists(d):
            mine.add(os.path.realpath(d))
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if mine and os.path.realpath(dev) not in mine:
            continue
        if glob.glob(dev + "/pp_dpm_sclk"):
            pw = glob.glob(dev + "/hwmon/hwmon*/power1_average") + glob.glob(dev + "/hwmon/hwmon*/power1_input")
            out.append((dev + "/pp_dpm_sclk", dev + "/pp_dpm_mclk", (pw[0] if pw else None)))
    return out


_PATHS = _sysfs_paths()


def _cur_level(path):
    for line in open(path):
        if line.rstrip().endswith("*"):
            m = re.search(r"(\d+)Mhz", line)
            if m:
                return int(m.group(1))
    return None


def _sample_card(paths):
    g = {}
    v = _cur_level(paths[0])
    if v is not None:
        g["sclk"] = v
    v = _cur_level(paths[1])
    if v is not None:
        g["mclk"] = v
    if paths[2]:
        g["power"] = int(open(paths[2]).read()) / 1e6
    return g


def sample():
    if _PATHS:                                                   # sysfs: hundreds of samples per second
        try:
            cards = [_sample_card(p) for p in _PATHS]
            if any(cards):
                return {"cards": cards}
        except Exception:                                        # noqa: BLE001
            pass
    try:
        out = subprocess.run(["rocm-smi", "-d", "0", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=5).stdout
        d = json.loads(out)
        card = d[sorted(d)[0]]
        g = {}
        for k, v in card.items():
            m = re.search(r"\((\d+)Mhz\)", str(v))
            if "sclk" in k and m:
                g["sclk"] = int(m.group(1))
            elif "mclk" in k and m:
                g["mclk"] = int(m.group(1))
            elif "ower" in k and "(W)" in k:
                try:
                    g["power"] = float(v)
                except ValueError:
                    pass
        return g
    except Exception:                                            # noqa: BLE001
        return {}


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    rows, stop = [], threading.Event()

    def loop():
        while not stop.is_set():
            g = sample()
            if g:
                rows.append(g)
            time.sleep(0.01)
    th = threading.Thread(target=loop, daemon=True)
    t0 = time.time()
    th.start()
    rc = subprocess.call(cmd)
    stop.set()
    th.join(timeout=6)
    with open(out_path, "a") as f:
        line = {"cmd": " ".join(cmd)[-160:], "seconds": round(time.time() - t0, 1), "samples": len(rows)}
        # only the samples taken while the GPU was busy say anything about the run: shader clock within 20 % of the highest seen
        if rows and "cards" in rows[0]:                          # keep the card that was busy: highest shader clock seen
            ncard = len(rows[0]["cards"])
            best = max(range(ncard), key=lambda i: max((r["cards"][i].get("sclk", 0) for r in rows if i < len(r["cards"])), default=0))
            line["card"] = best
            line["cards"] = ncard
            rows = [r["cards"][best] for r in rows if best < len(r["cards"])]
        if rows and any("sclk" in r for r in rows):
            top = max(r.get("sclk", 0) for r in rows)
            hot = [r for r in rows if r.get("sclk", 0) >= 0.8 * top]
            line["busy_samples"] = len(hot)
            for key in ("sclk", "mclk", "power"):
                v = [r[key] for r in hot if key in r]
                if v:
                    line[key] = {"median": statistics.median(v), "min": min(v), "max": max(v)}
        f.write(json.dumps(line) + "\n")
    return rc


if __name__ == "__main__":
    sys.exit(main())
