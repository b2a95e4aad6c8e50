"""`gbrs` command line for the two MI355X subcommands: same flags, defaults and log-and-continue
error behaviour as gbrs/commands.py:108-183 (`quantify`) and :153-183 (`reconstruct`).  argparse
instead of Typer so the GPU box needs nothing beyond the standard library."""
from __future__ import annotations

import argparse
import logging
import os
import sys


def configure_logging(verbose: int):
    """WARNING by default, level 19 with -v, DEBUG with -vv (utils.py:83-88)."""
    logger = logging.getLogger('gbrs')
    if not logger.handlers:
        h = logging.StreamHandler()
        h.setFormatter(logging.Formatter('[gbrs] %(message)s' if not os.getenv('GBRS_APP_DEBUG')
                                         else '[gbrs debug] %(levelname)s %(pathname)s:%(lineno)d %(message)s'))
        logger.addHandler(h)
    logger.setLevel(logging.WARNING if verbose == 0 else (19 if verbose == 1 else logging.DEBUG))
    return logger


def _existing(path):
    if not os.path.isfile(path):
        raise argparse.ArgumentTypeError(f"File '{path}' does not exist.")
    return os.path.realpath(path)


def build_parser():
    ap = argparse.ArgumentParser(prog='gbrs', description='GBRS numeric core on AMD MI355X')
    sub = ap.add_subparsers(dest='command', required=True)
    q = sub.add_parser('quantify', help='quantify allele-specific expressions')
    q.add_argument('-i', '--alignment-file', required=True, type=_existing)
    q.add_argument('-g', '--group-file', type=_existing, default=None)
    q.add_argument('-L', '--length-file', type=_existing, default=None)
    q.add_argument('-G', '--genotype', dest='genotype_file', type=_existing, default=None)
    q.add_argument('-o', '--outbase', default='gbrs.quantified')
    q.add_argument('-M', '--multiread-model', type=int, default=4)
    q.add_argument('-p', '--pseudocount', type=float, default=0.0)
    q.add_argument('-m', '--max-iters', type=int, default=999)
    q.add_argument('-t', '--tolerance', type=float, default=0.0001)
    q.add_argument('-a', '--report-alignment-counts', action='store_true')
    q.add_argument('-w', '--report-posterior', action='store_true')
    q.add_argument('-v', '--verbose', action='count', default=0)
    q.add_argument('--device', type=int, default=0, help='HIP device ordinal (extension)')
    q.add_argument('--merge-identical-rows', action='store_true',
                   help='merge identical reads into weighted rows on the device (extension)')
    r = sub.add_parser('reconstruct', help='reconstruct the genome based upon gene-level TPM quantities')
    r.add_argument('-e', '--expr-file', dest='expression_file', required=True, type=_existing)
    r.add_argument('-t', '--tprob-file', required=True, type=_existing)
    r.add_argument('-x', '--avec-file', type=_existing, default=None)
    r.add_argument('-g', '--gpos-file', type=_existing, default=None)
    r.add_argument('-c', '--expr-threshold', type=float, default=1.5)
    r.add_argument('-s', '--sigma', type=float, default=0.12)
    r.add_argument('-o', '--outbase', default=None)
    r.add_argument('-v', '--verbose', action='count', default=0)
    r.add_argument('--device', type=int, default=0, help='HIP device ordinal (extension)')
    it = sub.add_parser('interpolate', help='interpolate probability on a decently-spaced grid')
    it.add_argument('-i', '--genoprob-file', required=True, type=_existing)
    it.add_argument('-g', '--grid-file', type=_existing, default=None)
    it.add_argument('-p', '--gpos-file', type=_existing, default=None)
    it.add_argument('-o', '--output', dest='output_file', default=None)
    it.add_argument('-v', '--verbose', action='count', default=0)
    it.add_argument('--device', type=int, default=0)
    ex = sub.add_parser('export', help='export to GBRS quant format')
    ex.add_argument('-i', '--genoprob-file', required=True, type=_existing)
    ex.add_argument('-s', '--strains', action='append', required=True)
    ex.add_argument('-g', '--grid-file', type=_existing, default=None)
    ex.add_argument('-o', '--output', dest='output_file', default=None)
    ex.add_argument('-v', '--verbose', action='count', default=0)
    ex.add_argument('--device', type=int, default=0)
    cp = sub.add_parser('compress', help='compress EMASE format alignment incidence matrix')
    cp.add_argument('-i', '--emase-file', dest='emase_files', action='append', required=True)
    cp.add_argument('-o', '--output', dest='output_file', required=True)
    cp.add_argument('-c', '--comp-lib', default='zlib')
    cp.add_argument('-v', '--verbose', action='count', default=0)
    cp.add_argument('--device', type=int, default=0)
    wk = sub.add_parser('worker', help='(extension) quantify -> reconstruct -> quantify -G of many samples in one resident process')
    wk.add_argument('--jobs', required=True, type=_existing, help='JSON list of samples, see gbrs_amd/worker.py')
    wk.add_argument('-v', '--verbose', action='count', default=0)
    wk.add_argument('--device', type=int, default=0)
    wk.add_argument('--devices', default=None,
                    help='comma-separated HIP device ordinals, or "all": one resident worker process per entry, the samples '
                         'dealt round-robin (BASELINE configs[3]: 8 samples, one per GPU; replicas only, no collective)')
    return ap


def _write_stage_times(stages, error):
    """GBRS_STAGE_TIMES=<path> (measurement aid, scripts/e2e_bench.py): the wall-clock seconds of the
    driver's stages as JSON; with GBRS_T0=<time.time() of the launcher> also the start-up time up to
    main()."""
    path = os.getenv('GBRS_STAGE_TIMES')
    if not path:
        return
    import json
    if error is not None:
        stages['error'] = f'{type(error).__name__}: {error}'
    with open(path, 'w') as fh:
        json.dump(stages, fh)


def main(argv=None) -> int:
    import time
    t_main = time.time()
    stages = {}
    if os.getenv('GBRS_T0'):
        stages['startup'] = t_main - float(os.environ['GBRS_T0'])
    args = build_parser().parse_args(argv)
    logger = configure_logging(args.verbose)
    logger.debug(args.command)
    failure = None
    # as in the reference, failures are logged and the exit code stays 0 (commands.py:146-150)
    try:
        if args.command == 'quantify':
            if args.multiread_model not in (1, 2, 3, 4):
                raise RuntimeError('-M, --multiread-model must be one of 1, 2, 3, or 4')
            from .quantify import quantify
            quantify(alignment_file=args.alignment_file, group_file=args.group_file,
                     length_file=args.length_file, genotype_file=args.genotype_file, outbase=args.outbase,
                     multiread_model=args.multiread_model, pseudocount=args.pseudocount,
                     max_iters=args.max_iters, tolerance=args.tolerance,
                     report_alignment_counts=args.report_alignment_counts,
                     report_posterior=args.report_posterior, device=args.device,
                     merge_identical_rows=args.merge_identical_rows, stage_times=stages,
                     one_shot=True)         # the command builds one handle and exits: GBRS_EM_ONE_SHOT
        elif args.command == 'worker':
            from .worker import main as worker_main
            devices = None
            if args.devices:
                if args.devices == 'all':
                    from . import _lib
                    devices = list(range(max(1, int(_lib.load().gbrs_device_count()))))
                else:
                    devices = [int(x) for x in args.devices.split(',') if x.strip() != '']
            worker_main(args.jobs, device=args.device, devices=devices)
        elif args.command == 'compress':
            from .compress import compress
            files = [f for x in args.emase_files for f in x.split(',')]
            compress(emase_files=files, output_file=args.output_file, comp_lib=args.comp_lib, device=args.device)
        elif args.command == 'interpolate':
            from .postproc import interpolate
            interpolate(genoprob_file=args.genoprob_file, grid_file=args.grid_file, gpos_file=args.gpos_file,
                        output_file=args.output_file, device=args.device)
        elif args.command == 'export':
            from .postproc import export
            strains = [s for x in args.strains for s in x.split(',')]
            export(genoprob_file=args.genoprob_file, strains=strains, grid_file=args.grid_file,
                   output_file=args.output_file, device=args.device)
        else:
            from .hmm import reconstruct
            reconstruct(expression_file=args.expression_file, tprob_file=args.tprob_file,
                        avec_file=args.avec_file, gpos_file=args.gpos_file,
                        expr_threshold=args.expr_threshold, sigma=args.sigma, outbase=args.outbase,
                        device=args.device, stage_times=stages)
    except Exception as e:   # noqa: BLE001 - mirror of the reference's catch-all
        failure = e
        if logger.level == logging.DEBUG:
            logger.exception(e)
        else:
            logger.error(e)
    stages['main'] = time.time() - t_main
    _write_stage_times(stages, failure)
    return 0


def run() -> None:
    """Process entry point (`gbrs` console script, `python -m gbrs_amd`).  Everything the command leaves behind is
    on disk when main() returns, so the process ends with os._exit: the interpreter's and the HIP runtime's
    orderly teardown (module finalisers, unloading the code objects, releasing gigabytes of host arrays page by
    page) is ~0.1 s nobody waits for.  GBRS_ORDERLY_EXIT=1 keeps the ordinary exit."""
    code = main()
    if os.getenv('GBRS_ORDERLY_EXIT'):
        sys.exit(code)
    logging.shutdown()
    sys.stdout.flush()
    sys.stderr.flush()
    os._exit(code)


if __name__ == '__main__':
    run()
