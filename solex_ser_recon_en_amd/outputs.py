"""Output encoders and diagnostics, kept off the critical path.

The reference writes PNG/FITS products and three matplotlib diagnostics inline
(solex_util.py:263-273, 482-488, 556-587; ellipse_to_circle.py:316-341); each 400-dpi
plot costs seconds.  Here every file write is a background task (device -> host copy +
encode + write), so the GPU pipeline of the next disk / next file proceeds meanwhile:
image encoders run on a small pool (byte swaps, deflate, CRC and file writes release the
GIL; four 8 MB PNGs per file would otherwise take longer than decoding the next file),
the matplotlib diagnostics on one thread of their own.  A folder of scans with the default flags is then bound by
that thread (three 300-400 dpi figures, about 0.4 s each): SHG_PLOT_PROCESSES=N draws them in N spawned worker
processes instead (they import numpy and matplotlib only; images are copied to the host first; as with any
spawned pool, a calling script needs the usual `if __name__ == '__main__':` guard).  flush() re-raises the first failure, so a failed write still
stops the batch the way an exception in the reference's worker does (Solex_recon.py:42).
"""
import atexit
import multiprocessing
import os
import threading
from concurrent.futures import ProcessPoolExecutor, ThreadPoolExecutor

import numpy as np

from . import png_io

_lock = threading.Lock()
_pools = {}
_pending = []
ENCODER_THREADS = 4
_plot_procs = None


def _plot_processes():
    try:
        return max(0, int(os.environ.get('SHG_PLOT_PROCESSES', '0')))
    except ValueError:
        return 0


def _host_args(args):
    """Device images -> NumPy before they cross a process boundary."""
    return tuple(np.asarray(a) if hasattr(a, '__array__') and not isinstance(a, np.ndarray) else a for a in args)


def _plot_in_worker(fn, args):
    global _plot_procs
    host = _host_args(args)
    with _lock:
        if _plot_procs is None:
            _plot_procs = ProcessPoolExecutor(max_workers=_plot_processes(), mp_context=multiprocessing.get_context('spawn'))
            atexit.register(_plot_procs.shutdown)
        _pending.append(_plot_procs.submit(fn, *host))


synchronous = False          # tests can force inline execution


def _submitting_device():
    try:
        from .device import default_device
        return default_device()                             # the current device is thread-local: ask on the submitting thread
    except Exception:      # noqa: BLE001 -- encoders also serve host-only callers
        return None


def _off_the_scan_cores(device):
    """Encoders deflate and byte-swap megabytes: keep them off the cores the scan workers share (device.cpu_plan)."""
    if device is None:
        return
    try:
        from .device import bind_thread
        bind_thread('io', device)
    except Exception:      # noqa: BLE001 -- odd topology: stay where we are
        pass


def _stream_event(args):
    """The images were produced on the submitting thread's HIP stream; the encoder threads copy them to the host on
    their own (default) stream, so a task first waits for an event recorded behind the producing kernels."""
    if not any(hasattr(a, 't') and getattr(a.t, 'is_cuda', False) for a in args):
        return None
    import torch
    done = torch.cuda.Event()
    done.record()
    return done


def _run_after(done, fn, args):
    if done is not None:
        done.synchronize()
    return fn(*args)


def submit(fn, *args):
    if synchronous:
        fn(*args)
        return
    kind = 'plot' if getattr(fn, '__name__', '').startswith('plot_') else 'encode'
    done = _stream_event(args)
    with _lock:
        if kind not in _pools:
            _pools[kind] = ThreadPoolExecutor(max_workers=1 if kind == 'plot' else ENCODER_THREADS,
                                              thread_name_prefix='shg-' + kind, initializer=_off_the_scan_cores,
                                              initargs=(_submitting_device(),))
        if kind == 'plot' and _plot_processes() > 0:
            # the plot thread only copies the images to the host and hands the figure to a worker process
            _pending.append(_pools[kind].submit(_run_after, done, _plot_in_worker, (fn, args)))
        else:
            _pending.append(_pools[kind].submit(_run_after, done, fn, args))


def flush():
    err = None
    while True:
        with _lock:
            todo = list(_pending)
            del _pending[:]
        if not todo:
            break                    # (a plot task may have queued its worker-process future while we waited)
        for fut in todo:
            try:
                fut.result()
            except Exception as e:      # noqa: BLE001 -- keep draining, report the first
                err = err or e
    if err is not None:
        raise err


def write_png16(path, img):
    png_io.write_png(path, np.asarray(img), 0)


# ---- diagnostics (matplotlib Agg figures; same content as the reference's plots) ----------
def _figure():
    import matplotlib
    matplotlib.use('Agg', force=False)
    import matplotlib.figure
    return matplotlib.figure.Figure()


def plot_spectral_line(path, mean_img, xs, ys, curve, ih, stride):
    import matplotlib.pyplot
    fig = _figure()
    ax = fig.add_subplot(1, 1, 1)
    ax.imshow(np.asarray(mean_img), cmap=matplotlib.pyplot.cm.gray)
    ax.plot(xs[::stride], ys[::stride], 'rx', label='line detection')
    ax.plot(curve, np.arange(ih), label='polynomial fit')
    ax.legend(loc='center left', bbox_to_anchor=(1, 0.5))
    ax.set_aspect(0.1)
    fig.tight_layout()
    fig.savefig(path, dpi=400)


def plot_transversalium(path, c):
    fig = _figure()
    ax = fig.add_subplot(1, 1, 1)
    ax.plot(c)
    ax.set_xlabel('y')
    ax.set_ylabel('transversalium correction factor')
    fig.savefig(path, dpi=300)


def plot_ellipse_fit(path, image, fix_img, raw_X, X_f, ellipse_points, borders):
    import matplotlib.pyplot
    gray = matplotlib.pyplot.cm.gray
    image = np.asarray(image)
    fig = _figure()
    ax = [[fig.add_subplot(2, 2, 1), fig.add_subplot(2, 2, 2)], [fig.add_subplot(2, 2, 3), fig.add_subplot(2, 2, 4)]]
    fig.tight_layout()
    ax[0][0].imshow(image, cmap=gray)
    ax[0][0].set_title('uncorrected image', fontsize=11)
    ax[0][0].set_aspect('equal')
    ax[0][1].set_aspect('equal')
    ax[0][1].imshow(image, cmap=gray)
    ax[0][1].plot(raw_X[:, 1], raw_X[:, 0], 'ro', label='edge detection')
    ax[0][1].legend(prop={'size': 6})
    ax[1][1].set_aspect('equal')
    ax[1][1].plot(X_f[:, 1], X_f[:, 0], 'ro', label='filtered edges')
    ax[1][1].plot(ellipse_points[:, 1], ellipse_points[:, 0], color='b', label='ellipse fit')
    ax[1][1].set_ylim([image.shape[0], 0])
    ax[1][1].legend(prop={'size': 6})
    ax[1][0].set_aspect('equal')
    ax[1][0].imshow(np.asarray(fix_img), cmap=gray)
    ax[1][0].axhline(y=borders[1])
    ax[1][0].axhline(y=borders[3])
    ax[1][0].axvline(x=borders[0])
    ax[1][0].axvline(x=borders[2])
    ax[1][0].set_title('geometrically corrected image', fontsize=11)
    fig.savefig(path, dpi=300)
