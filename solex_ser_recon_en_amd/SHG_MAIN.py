"""Front door: `python -m solex_ser_recon_en_amd.SHG_MAIN [-flags] files...`.

The command-line branch of the reference's SHG_MAIN.py (:218-223, :248) with its option
schema (:41-68), task list (precheck_files :98-132), folder glob (:154-159) and error
handling (handle_files :134-143: print the traceback, never re-raise).  The GUI, the ini
file and the language packs are out of scope.  Under `torchrun` (one process per GPU) a
single file is frame-sharded over the ranks, several files are dealt one per rank.
"""
import os
import sys
import traceback

from . import CLI_handler, Solex_recon

options = {
    'language': 'English',
    'shift': [0],
    'flag_display': False,
    'ratio_fixe': None,
    'slant_fix': None,
    'save_fit': False,
    'clahe_only': False,
    'protus_only': False,
    'disk_display': True,
    'delta_radius': 0,
    'crop_width_square': False,
    'transversalium': True,
    'stubborn_transversalium': False,
    'trans_strength': 301,
    'img_rotate': 0,
    'flip_x': False,
    'workDir': '',
    'fixed_width': None,
    'output_dir': '',
    'input_dir': '',
    'specDir': '',
    'selected_mode': 'File input mode',
    'continuous_detect_mode': False,
    'dispersion': 0.05,
    'ellipse_fit_shift': 10,
    'de-vignette': False,
}


def default_options():
    return {k: (list(v) if isinstance(v, list) else v) for k, v in options.items()}


def _openable(path):
    try:
        with open(path, 'rb'):
            return True
    except Exception:
        traceback.print_exc()
        return False


def precheck_files(serfiles, options):
    """The task list of SHG_MAIN.py:98-132: one (file, copy of the options) per file that has a name and can be opened;
    the rest are reported and skipped.  `tempo` is the GUI's display pause (unused here, kept in the schema)."""
    options['tempo'] = 30000 if len(serfiles) == 1 else 5000
    tasks = []
    for path in serfiles:
        print(path)
        if path == '':
            print("ERROR filename empty")
        elif os.path.basename(path) == '':
            print('filename ERROR : ', path)
        elif not _openable(path):
            print('ERROR opening file : ', path)
        else:
            tasks.append((path, options.copy()))
    return tasks


def handle_files(files, options, flag_command_line=False):
    good_tasks = precheck_files(files, options)
    try:
        # several files under torch.distributed: one file per rank (folder mode) unless SHG_DISTRIBUTE=frames asks for every
        # file's frames to be sharded over the ranks (scans longer than one GPU's share of the link / of HBM)
        Solex_recon.solex_do_work(good_tasks, flag_command_line, distribute=os.environ.get('SHG_DISTRIBUTE', 'auto'))
        return True
    except Exception:
        print('ERROR ENCOUNTERED')
        traceback.print_exc()
        return False


def scans_in(folder):
    """The SER / AVI scans of a folder, sorted (the reference's folder mode globs *.ser, SHG_MAIN.py:146-150)."""
    return sorted(os.path.join(folder, f) for f in os.listdir(folder)
                  if f.rsplit('.', 1)[-1].upper() in ('SER', 'AVI') and os.path.isfile(os.path.join(folder, f)))


def handle_folder(options):
    return handle_files(scans_in(options['input_dir']), options, True)


def _init_distributed():
    if int(os.environ.get('WORLD_SIZE', '1')) <= 1:
        return
    import torch
    import torch.distributed as td
    backend = os.environ.get('SHG_DIST_BACKEND', 'nccl')     # 'gloo': functional runs with several ranks on one GPU
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    torch.cuda.set_device(local_rank if backend == 'nccl' else local_rank % max(torch.cuda.device_count(), 1))
    td.init_process_group(backend)


def _shutdown_distributed():
    import torch.distributed as td
    if td.is_available() and td.is_initialized():
        td.destroy_process_group()


def main(argv=None):
    opts = default_options()
    argv = sys.argv[1:] if argv is None else list(argv)
    # a folder argument stands for the scans inside it (one file per GPU under torch.distributed)
    argv = [f for a in argv for f in (scans_in(a) if not a.startswith('-') and os.path.isdir(a) else [a])]
    serfiles = CLI_handler.handle_CLI(opts, argv)
    if not serfiles:
        print(CLI_handler.usage())
        return 1
    _init_distributed()
    # everything imported and set up so far lives as long as the process: keep the cyclic collector's full passes (85 ms
    # with torch loaded, every scan worker stopped) away from it
    import gc
    gc.freeze()
    try:
        return 0 if handle_files(serfiles, opts, flag_command_line=True) else 1
    finally:
        _shutdown_distributed()


if __name__ == '__main__':
    sys.exit(main())
