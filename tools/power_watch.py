"""Run a command and sample rocm-smi beside it (socket power, sclk, junction temperature): is a sustained run power-managed?
usage: python tools/power_watch.py [--period 0.2] -- <command ...>     (the child is started as a plain subprocess; this
process never touches HIP).  Prints the child's output, then one JSON line with the samples' summary.
       python tools/power_watch.py [--period 0.2] --until-eof          samples until its stdin closes, then prints ONE JSON
line {"cap": ..., "samples": [[unix time, W, sclk MHz, junction C], ...]} (bench.py starts it this way, before it touches HIP)."""
import json
import os
import subprocess
import sys
import threading
import time


def _rocm_smi_cmd():
    """rocm-smi is a `#!/usr/bin/env python3` script: run it with THIS interpreter, never through the env shebang (ADVICE r3:
    one exec hop fewer per sample, and nothing that re-execs under a preloaded tool library)"""
    import shutil
    path = shutil.which("rocm-smi") or "/opt/rocm/bin/rocm-smi"
    real = os.path.realpath(path)
    try:
        with open(real, "rb") as f:
            if f.read(2) == b"#!" and b"python" in f.readline():
                return [sys.executable, real]
    except OSError:
        pass
    return [path]


def _sysfs_sensors():
    """-> dict of sysfs paths of device 0's power / clock / temperature sensors, or None where the driver does not expose them"""
    import glob
    for hw in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        pw = next((p for p in (hw + "/power1_average", hw + "/power1_input") if os.path.exists(p)), None)
        fq = hw + "/freq1_input"
        if pw and os.path.exists(fq):
            tj = next((p for p in (hw + "/temp2_input", hw + "/temp1_input") if os.path.exists(p)), None)
            cap = hw + "/power1_cap"
            return {"power": pw, "freq": fq, "temp": tj, "cap": cap if os.path.exists(cap) else None}
    return None


_SYSFS = None


def sample():
    global _SYSFS
    if _SYSFS is None:
        _SYSFS = _sysfs_sensors() or False
    if _SYSFS:
        rd = lambda p: float(open(p).read().strip())   # noqa: E731
        return {"power_w": rd(_SYSFS["power"]) / 1e6, "sclk_mhz": rd(_SYSFS["freq"]) / 1e6,
                "temp_c": rd(_SYSFS["temp"]) / 1e3 if _SYSFS["temp"] else 0.0}
    r = subprocess.run(_rocm_smi_cmd() + ["-d", "0", "--showpower", "--showclocks", "--showtemp", "--json"], capture_output=True,
                       text=True, timeout=5)
    d = json.loads(r.stdout)["card0"]
    num = lambda s: float("".join(ch for ch in s if ch.isdigit() or ch == "."))   # noqa: E731
    return {"power_w": num(d["Current Socket Graphics Package Power (W)"]), "sclk_mhz": num(d["sclk clock speed:"]),
            "temp_c": num(d["Temperature (Sensor junction) (C)"])}


def main():
    argv = sys.argv[1:]
    period = 0.2
    if argv and argv[0] == "--period":
        period = float(argv[1])
        argv = argv[2:]
    if argv and argv[0] == "--until-eof":
        cap_w = None
        sysfs = _sysfs_sensors()
        try:
            if sysfs and sysfs["cap"]:
                cap_w = float(open(sysfs["cap"]).read().strip()) / 1e6
            else:
                cap = subprocess.run(_rocm_smi_cmd() + ["-d", "0", "--showmaxpower", "--json"], capture_output=True, text=True,
                                     timeout=10).stdout
                cap_w = float(json.loads(cap)["card0"]["Max Graphics Package Power (W)"])
        except Exception:  # noqa: BLE001
            cap_w = None
        rows, stop = [], threading.Event()

        def loop_eof():
            while not stop.is_set():
                try:
                    s_ = sample()
                    rows.append([round(time.time(), 3), s_["power_w"], s_["sclk_mhz"], s_["temp_c"]])
                except Exception:  # noqa: BLE001
                    pass
                stop.wait(period)

        th = threading.Thread(target=loop_eof, daemon=True)
        th.start()
        sys.stdin.read()            # returns when the parent closes the pipe (or dies)
        stop.set()
        th.join(timeout=10)
        print(json.dumps({"cap_w": cap_w, "samples": rows, "source": "sysfs hwmon" if sysfs else "rocm-smi"}), flush=True)
        return
    if argv and argv[0] == "--":
        argv = argv[1:]
    cap = subprocess.run(_rocm_smi_cmd() + ["-d", "0", "--showmaxpower", "--json"], capture_output=True, text=True).stdout.strip()
    rows, stop = [], threading.Event()

    def loop():
        t0 = time.time()
        while not stop.is_set():
            try:
                s = sample()
                s["t"] = round(time.time() - t0, 2)
                rows.append(s)
            except Exception as ex:  # noqa: BLE001
                rows.append({"t": round(time.time() - t0, 2), "error": repr(ex)[:100]})
            time.sleep(period)

    th = threading.Thread(target=loop)
    th.start()
    rc = subprocess.run(argv).returncode
    stop.set()
    th.join()
    good = [r for r in rows if "power_w" in r]
    busy = [r for r in good if r["power_w"] > 500]
    summ = {"max_power_cap": cap[:200], "samples": len(good), "busy_samples(>500W)": len(busy)}
    if busy:
        p = sorted(r["power_w"] for r in busy)
        c = sorted(r["sclk_mhz"] for r in busy)
        summ.update({"power_w_median": p[len(p) // 2], "power_w_max": p[-1], "sclk_mhz_median": c[len(c) // 2], "sclk_mhz_min": c[0],
                     "sclk_mhz_max": c[-1], "temp_c_max": max(r["temp_c"] for r in busy)})
    summ["trace(t,power,sclk)"] = [(r["t"], r["power_w"], r["sclk_mhz"]) for r in good][:400]
    print(json.dumps({"power_watch": summ}))
    sys.exit(rc)


if __name__ == "__main__":
    main()
