"""One resident process per GPU for many samples: `gbrs quantify` -> `gbrs reconstruct` -> `gbrs quantify -G` of every
sample without starting the interpreter and the HIP runtime three times per sample, without reading the sample's alignment
file twice (the multiway and the diploid pass read the same `.h5`; gbrs/emase_utils.py:180-332 is written as two commands)
and without reloading the sample-independent tables of `reconstruct` (transition probabilities, specificity blocks, gene
order; gbrs_utils.py:420-447) for every sample.  The three steps are the very functions the commands call
(gbrs_amd.quantify.quantify, gbrs_amd.hmm.reconstruct): same files out.

    python -m gbrs_amd worker --jobs jobs.json [--device 0]
    python -m gbrs_amd worker --jobs jobs.json --devices 0,1,2,3,4,5,6,7     # one worker process per GPU, samples dealt round-robin

jobs.json: a list of objects {"alignment_file", "outbase", "tprob_file", and optionally "group_file", "length_file",
"avec_file", "gpos_file", "expr_threshold", "sigma", "pseudocount", "max_iters", "tolerance", "diploid": true|false};
the stage times of every sample are printed as one JSON line per sample.
"""
from __future__ import annotations

import json
import logging
import os
import time

logger = logging.getLogger('gbrs')


class SampleWorker:
    """The resident state: the reconstruct context of the last table set, nothing else (a sample's alignment file is
    held for the two quantify passes of that sample only: 1.5 GB of index arrays at DO size)."""

    def __init__(self, device: int = 0):
        self.device = device
        self._ctx = None
        self._ctx_key = None

    def _context(self, job):
        from .hmm import ReconstructContext
        key = (job['tprob_file'], job.get('avec_file'), job.get('gpos_file'))
        if key != self._ctx_key:
            if self._ctx is not None:
                self._ctx.close()
            self._ctx = ReconstructContext(job['tprob_file'], job.get('avec_file'), job.get('gpos_file'), self.device)
            self._ctx_key = key
        return self._ctx

    def process(self, job: dict) -> dict:
        """One sample: multiway quantification, genome reconstruction from its gene TPMs, diploid quantification under
        the reconstructed genotypes.  Returns the stage times (seconds)."""
        from .alignment import load_alignment
        from .em import read_length_file
        from .hmm import reconstruct
        from .quantify import quantify
        clock = time.perf_counter
        t_all = clock()
        out = {'outbase': job['outbase']}
        common = dict(group_file=job.get('group_file'), length_file=job.get('length_file'),
                      multiread_model=4, pseudocount=job.get('pseudocount', 0.0), max_iters=job.get('max_iters', 999),
                      tolerance=job.get('tolerance', 0.0001), device=self.device)
        # the sample's alignments and lengths, once for both passes
        t0 = clock()
        from concurrent.futures import ThreadPoolExecutor
        side = ThreadPoolExecutor(max_workers=1)
        pending = []

        def names_known(apm):
            if job.get('length_file'):
                pending.append(side.submit(read_length_file, apm, job['length_file'], 100))
        aln = load_alignment(job['alignment_file'], grpfile=job.get('group_file'), on_names=names_known)
        lengths = pending[0].result() if pending else None
        side.shutdown(wait=False)
        out['load_alignment'] = clock() - t0
        st = {}
        t0 = clock()
        quantify(alignment_file=job['alignment_file'], outbase=job['outbase'], stage_times=st, alignment=aln,
                 target_lengths=lengths, **common)
        out['quantify'] = dict(st, wall=clock() - t0)
        st = {}
        t0 = clock()
        reconstruct(expression_file=f"{job['outbase']}.multiway.genes.tpm", tprob_file=job['tprob_file'],
                    avec_file=job.get('avec_file'), gpos_file=job.get('gpos_file'),
                    expr_threshold=job.get('expr_threshold', 1.5), sigma=job.get('sigma', 0.12),
                    outbase=job['outbase'], device=self.device, stage_times=st, context=self._context(job))
        out['reconstruct'] = dict(st, wall=clock() - t0)
        if job.get('diploid', True):
            st = {}
            t0 = clock()
            quantify(alignment_file=job['alignment_file'], genotype_file=f"{job['outbase']}.genotypes.tsv",
                     outbase=job['outbase'], stage_times=st, alignment=aln, target_lengths=lengths, **common)
            out['quantify_diploid'] = dict(st, wall=clock() - t0)
        out['wall'] = clock() - t_all
        return out

    def close(self):
        if self._ctx is not None:
            self._ctx.close()
            self._ctx = None


def run_jobs(jobs, device: int = 0, emit=print):
    """Process the jobs in order on one device; failures are logged and the next sample goes on (the commands'
    log-and-continue behaviour, gbrs/commands.py:146-150)."""
    w = SampleWorker(device)
    t0 = time.perf_counter()
    done = []
    for job in jobs:
        try:
            res = w.process(job)
        except Exception as e:   # noqa: BLE001 - one bad sample does not end the queue
            logger.error(e)
            res = {'outbase': job.get('outbase'), 'error': f'{type(e).__name__}: {e}'}
        done.append(res)
        if emit:
            emit(json.dumps(res), flush=True)
    w.close()
    return done, time.perf_counter() - t0


def run_jobs_on_devices(jobs, devices, emit=print):
    """BASELINE configs[3] as a product path: the samples of `jobs` spread over several GPUs, ONE resident worker process
    per entry of `devices` (a device may be listed twice: two workers share it), no collective and no torch - samples are
    independent (replicas only).  Jobs are handed out round-robin (sample k to worker k mod N), every child is
    `python -m gbrs_amd worker --jobs <its share> --device d`; its per-sample JSON lines are relayed as they come."""
    import subprocess
    import sys
    import tempfile
    import threading
    n = len(devices)
    shares = [[j for k, j in enumerate(jobs) if k % n == w] for w in range(n)]
    tmp = tempfile.mkdtemp(prefix='gbrs_worker_')
    procs, results, lock = [], [], threading.Lock()
    t0 = time.perf_counter()
    for w, (dev, share) in enumerate(zip(devices, shares)):
        if not share:
            continue
        jf = os.path.join(tmp, f'jobs_{w}.json')
        with open(jf, 'w') as fh:
            json.dump(share, fh)
        procs.append((w, dev, subprocess.Popen([sys.executable, '-m', 'gbrs_amd', 'worker', '--jobs', jf, '--device', str(dev)],
                                               stdout=subprocess.PIPE, text=True)))

    def relay(w, dev, proc):
        for line in proc.stdout:
            if not line.startswith('{'):
                continue                      # (EMfactory.run's iteration table)
            rec = json.loads(line)
            if 'summary' in rec:
                continue
            rec['worker'], rec['device'] = w, dev
            with lock:
                results.append(rec)
                if emit:
                    emit(json.dumps(rec), flush=True)
        proc.wait()
    threads = [threading.Thread(target=relay, args=p) for p in procs]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    import shutil
    shutil.rmtree(tmp, ignore_errors=True)
    failed = [p.returncode for _, _, p in procs if p.returncode != 0]
    if failed:
        raise RuntimeError(f'{len(failed)} worker process(es) failed: exit codes {failed}')
    return results, time.perf_counter() - t0


def main(jobs_file: str, device: int = 0, devices=None) -> int:
    with open(jobs_file) as fh:
        jobs = json.load(fh)
    t_start = float(os.environ['GBRS_T0']) if os.getenv('GBRS_T0') else None
    if devices:
        done, seconds = run_jobs_on_devices(jobs, devices)
        summary = {'samples': len(done), 'failed': sum('error' in d for d in done) + (len(jobs) - len(done)),
                   'seconds': seconds, 'workers': len(devices), 'samples_per_s': len(done) / seconds if seconds > 0 else None}
        print(json.dumps({'summary': summary}), flush=True)
        return 0
    done, seconds = run_jobs(jobs, device)
    summary = {'samples': len(done), 'failed': sum('error' in d for d in done), 'seconds_in_worker': seconds}
    if t_start is not None:
        summary['seconds_since_launch'] = time.time() - t_start
    print(json.dumps({'summary': summary}), flush=True)
    return 0
