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
    import glob
    for dev in sorted(glob.glob("/sys/class/drm/card*/device")):
        if glob.glob(dev + "/pp_dpm_sclk"):
            pw = glob.glob(dev + "/hwmon/hwmon*/power1_average") + glob.glob(dev + "/hwmon/hwmon*/power1_input")
            return dev + "/pp_dpm_sclk", dev + "/pp_dpm_mclk", (pw[0] if pw else None)
    return None


_PATHS = _sysfs_paths()


def _cur_level(path):
    for line in open(path):
        if line.rstrip().endswith("*"):
            m = re.search(r"(\d+)Mhz", line)
            if m:
                return int(m.group(1))
    return None


def sample():
    if _PATHS:                                                   # sysfs: hundreds of samples per second
        try:
            g = {}
            v = _cur_level(_PATHS[0])
            if v is not None:
                g["sclk"] = v
            v = _cur_level(_PATHS[1])
            if v is not None:
                g["mclk"] = v
            if _PATHS[2]:
                g["power"] = int(open(_PATHS[2]).read()) / 1e6
            if g:
                return g
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
