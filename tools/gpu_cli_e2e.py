#!/usr/bin/env python3
"""Lease script: the metric as SURVEY 8(d) defines it -- wall time of the whole CLI run, file read -> .rl_bwt closed.

Writes the workload to a file (page cache), runs grlbwt_amd/bin/grlbwt on it, reports the CLI's own timing line, the
wall clock around the process, and checks the output file's md5 against the image of an in-HBM build of the same bytes.
  python3 tools/gpu_cli_e2e.py [sizes...]     sizes in {101MB, 1GB, 10GB}; results -> gpurun_out/cli_e2e.json
"""
import hashlib
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    import __graft_entry__ as g
    from grlbwt_amd import engine, workloads
    lib = g.build_hip()
    cli = g.build_cli()
    sizes = sys.argv[1:] or ["101MB", "1GB", "10GB"]
    tmp = os.environ.get("GRLBWT_E2E_TMP", "/tmp")
    dev = torch.device("cuda", 0)
    out = []
    for sz in sizes:
        if sz == "101MB":
            text = workloads.uniform_reads_torch(1000000, 100, seed=20260001, device=dev)
        elif sz == "1GB":
            text = workloads.sampled_reads_torch(6622517, 150, 33000000, seed=20260003, device=dev)
        else:
            text = workloads.sampled_reads_torch(66225166, 150, 330000000, seed=20260003, device=dev)
        torch.cuda.synchronize()
        fin, fout = os.path.join(tmp, "e2e_%s.txt" % sz), os.path.join(tmp, "e2e_%s.rl_bwt" % sz)
        # reference image (in-HBM build) md5, chunked download
        want, t_hbm, nb = None, 0.0, 0
        if os.environ.get("GRLBWT_E2E_NO_PARENT_BUILD"):      # the CLI runs on a device this process has not built on
            n = int(text.numel())
            with open(fin, "wb") as f:
                for a in range(0, n, 1 << 28):
                    f.write(text[a:a + (1 << 28)].cpu().numpy().tobytes())
        else:
          with engine.Context(0, 0, lib) as ctx:
              ctx.attach_device(text.data_ptr(), text.numel(), 1, keepalive=text)
              t0 = time.time()
              ctx.build()
              torch.cuda.synchronize()
              t_hbm = time.time() - t0
              nb, nr = ctx.result_size()
              from grlbwt_amd import dist as gdist
              img = gdist._view(ctx.result_device_ptr(), nb, dev)
              h = hashlib.md5()
              for a in range(0, nb, 1 << 28):
                  h.update(img[a:a + (1 << 28)].cpu().numpy().tobytes())
              want = h.hexdigest()
        n = int(text.numel())
        if want is not None:
            with open(fin, "wb") as f:
                for a in range(0, n, 1 << 28):
                    f.write(text[a:a + (1 << 28)].cpu().numpy().tobytes())
        if not os.environ.get("GRLBWT_E2E_KEEP_TEXT"):
            del text
        torch.cuda.empty_cache()
        runs = []
        for rep in range(int(os.environ.get("GRLBWT_E2E_REPS", "2"))):
            try:
                os.remove(fout)                  # a fresh output file (replacing an existing 8 GB one costs ~0.7 s inside rename())
            except OSError:
                pass
            time.sleep(float(os.environ.get("GRLBWT_E2E_PAUSE", "2")))
            t0 = time.time()
            p = subprocess.run([cli, fin, "-o", fout], capture_output=True, text=True)
            wall = time.time() - t0
            if p.returncode != 0:
                print(p.stdout[-2000:], p.stderr[-2000:], file=sys.stderr)
                raise SystemExit("CLI failed on " + sz)
            if os.environ.get("GRLBWT_E2E_VERBOSE"):
                keep = [l for l in p.stdout.splitlines() if l.startswith(("  Parsing round", "  Inducing", "grlbwt-timing")) or
                        (l.startswith("    Elapsed")) or "Elapsed time" in l and not l.startswith("     ")]
                print("\n".join(keep[-60:]), p.stderr[-2000:], file=sys.stderr, flush=True)
            m = re.search(r"grlbwt-timing: read\+upload ([\d.]+) s, build ([\d.]+) s, write ([\d.]+) s, total ([\d.]+) s", p.stdout)
            hh = hashlib.md5()
            with open(fout, "rb") as f:
                for blk in iter(lambda: f.read(1 << 26), b""):
                    hh.update(blk)
            runs.append({"wall_s": round(wall, 3), "read_upload_s": float(m.group(1)), "build_s": float(m.group(2)), "write_s": float(m.group(3)),
                         "total_s": float(m.group(4)), "MBps_wall": round(n / 1e6 / wall, 1), "md5_ok": (hh.hexdigest() == want) if want else None})
        labels = [l for l in ("Computing the dictionary of LMS phrases", "Creating the parse of the text", "Assembling the new BWT",
                              "Performing the induction from the previous BWT") if l in p.stdout]
        out.append({"size": sz, "input_bytes": n, "image_bytes": nb, "hbm_build_s_first": round(t_hbm, 3), "runs": runs, "stage_labels_found": len(labels)})
        print(json.dumps(out[-1]), flush=True)
        os.remove(fin)
        os.remove(fout)
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(out, open(os.path.join(ROOT, "gpurun_out", "cli_e2e.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
