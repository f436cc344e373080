"""The reference's pipeline callers around the estimator (SURVEY 8f-1,
``openpose/native/python/inference.py``): the step before the hot path (file -> BGR array ->
reshape to ``op_image_height x op_image_width``) and the step after it (copies of the results
handed to a tracker, CSV / image writers), with the reference's function names, argument
meaning, result layouts and error behaviour:

* :func:`extract_2dskeletons` (``inference.py:168-282``) -> ``{'status', 'data_prep_time', 'infer_time'}``
* :class:`ExtractSkeletonAndTrack` (``:22-165``): three threads joined by bounded queues, tuple
  layouts of ``:83-116``, ``:120-146``, ``:150-165``, sentinel shutdown
* :func:`extract_2dskeletons_online` / ``_offline`` / ``_and_track_offline`` (``:285-534``), :func:`main`
  (mode dispatch of ``:577-601``)

The tracker (``tracking/track.py``) is outside the path (out of scope); any object with
``update / no_measurement_predict_and_update / tracks / name`` can be passed in, and when none is
given ``tracking.track.Tracker`` is imported from the caller's own tree if it is importable.

Addition: :func:`extract_2dskeletons_offline_batched` runs the same folder walk with
``--posehip-batch`` frames per engine call (the reference is batch 1 throughout).
"""
import argparse
import os
from concurrent.futures import ThreadPoolExecutor
from queue import Queue
from threading import Thread
from typing import List, Optional, Union

import numpy as np

from .api_native import OpenPosePoseExtractor
from .api_native import PyOpenPoseNative as PYOP
from .args import get_parser
from .utils import Timer
from .utils_rs import prepare_save_paths, read_color_file, read_depth_file

_FAILED = {'status': False, 'data_prep_time': -1, 'infer_time': -1}


def _progress(iterable):
    try:
        from tqdm import tqdm
        return tqdm(iterable, dynamic_ncols=True)
    except ImportError:  # pragma: no cover
        class _Plain(list):
            def set_description(self, text):
                print(text)
        return _Plain(iterable)


def _write_image(path: str, bgr: np.ndarray) -> None:
    from PIL import Image
    Image.fromarray(np.ascontiguousarray(bgr[:, :, ::-1])).save(path)


class _FrameSkipped(Exception):
    """A frame that cannot go through the estimator: the cause and the warning lines the
    reference prints for it (``inference.py:208-215,237-247,271-278``)."""

    def __init__(self, cause, *lines):
        super().__init__(cause)
        self.cause, self.lines = cause, lines

    def report(self):
        print(self.cause)
        for line in self.lines:
            print(line)


def _fetch(src, reader):
    """A source given as a path goes through `reader`; an array is taken as it is."""
    if not isinstance(src, str):
        assert isinstance(src, np.ndarray)
        return src
    try:
        return reader(src)
    except Exception as cause:
        raise _FrameSkipped(cause, f"[WARN] : Error loading data, skipping {src}") from cause


def extract_2dskeletons(args: argparse.Namespace,
                        pose_extractor: OpenPosePoseExtractor,
                        color_src: Union[np.ndarray, str],
                        depth_src: Optional[Union[np.ndarray, str]] = None,
                        intr_mat: Optional[Union[np.ndarray, str]] = None,
                        skeleton_folder: Optional[str] = None,
                        skeleton_file_prefix: str = '',
                        enable_timer: bool = False) -> dict:
    """One frame through the estimator (``inference.py:168-282``).

    Returns ``{'status': bool, 'data_prep_time': s, 'infer_time': s}``; on any load, reshape or
    inference error prints a warning and returns status False with both times -1.  The times are
    None when ``enable_timer`` is False (the reference's Timer).

    The 3-D branch passes its arguments to ``predict_3d`` positionally exactly as the reference
    does (``:256-263``): the 2-D save path lands in ``depth_scale``.  That is the reference's
    behaviour (a frame with people then fails inside ``convert_to_3d`` and reports status False
    unless ``op_save_skel`` is off *and* nobody is found); it is kept, not repaired.
    """
    assert color_src is not None
    # second warning line of the reshape / inference failures: only a file has a name to print
    origin = (f"skipping {color_src}",) if isinstance(color_src, str) else ()
    lift = bool(args.op_extract_3d_skel)
    try:
        with Timer("data preparation", enable_timer, printout=False) as prep:
            paths = prepare_save_paths(skeleton_folder, skeleton_file_prefix, args.op_save_skel,
                                       args.op_save_skel_image, args.op_save_3d_skel)
            image = _fetch(color_src, read_color_file)
            depth = _fetch(depth_src, read_depth_file) if lift and depth_src is not None else None
            try:  # the estimator takes exactly H x W x 3
                image = image.reshape(args.op_image_height, args.op_image_width, 3)
            except Exception as cause:
                raise _FrameSkipped(cause, "[WARN] : Error in reshaping image...", *origin) from cause
        with Timer("infer", enable_timer, printout=False) as infer:
            csv_2d, csv_3d, jpg = paths['skel_save_path'], paths['3dskel_save_path'], paths['skel_image_save_path']
            try:
                if lift:
                    pose_extractor.predict_3d(image, depth, intr_mat, csv_2d, csv_3d, jpg)
                else:
                    pose_extractor.predict(image, depth, csv_2d, jpg)
            except Exception as cause:
                raise _FrameSkipped(cause, "[WARN] : Error in pose extraction...", *origin) from cause
    except _FrameSkipped as skipped:
        skipped.report()
        return dict(_FAILED)
    return {'status': True, 'data_prep_time': prep.duration, 'infer_time': infer.duration}


class ExtractSkeletonAndTrack:
    """Estimator -> tracker -> writer as three threads joined by queues of depth 10
    (``inference.py:22-165``).

    queue_input((color_path, depth_path, skeleton_folder, intr_mat, break_loop))
    queue_output() -> (filtered_skel, prep_time, infer_time, track_time, break_loop)

    ``break_process_loops()`` sends the sentinel through all three stages and waits for it.

    Every queue carries tuples whose LAST field is the stop flag; the layouts between the stages
    are the reference's (``:83-116``, ``:120-146``, ``:150-165``).  One pump (`_pump`) serves all
    three threads: it forwards the sentinel, padded to the next layout's width, and otherwise
    hands the tuple's fields to the stage function.
    """
    _MP_MESSAGE = "mp not supported for now, openpose inference seems to cause the mp to hang."

    def __init__(self, args, pose_extractor, tracker, enable_timer) -> None:
        self.args = args
        self.PE = pose_extractor
        self.TK = tracker
        self.enable_timer = enable_timer
        self._refuse_mp()
        self.infer_queue, self.track_queue, self.write_queue, self.final_queue = (Queue(maxsize=10) for _ in range(4))

    def _refuse_mp(self):
        if self.args.op_proc == 'mp':
            raise ValueError(self._MP_MESSAGE)

    def start(self):
        self._refuse_mp()
        self.p_extract = Thread(target=self.extract_skeletons, args=())
        self.p_track = Thread(target=self.track, args=())
        self.p_write = Thread(target=self.writeout, args=())
        for worker in (self.p_extract, self.p_track, self.p_write):
            worker.start()

    def queue_input(self, args):
        self.infer_queue.put(args)

    def queue_output(self):
        return self.final_queue.get()

    def break_process_loops(self):
        self.queue_input((None, None, None, None, True))
        self.queue_output()

    @staticmethod
    def _pump(inbox, outbox, out_width, stage):
        while True:
            *fields, stop = inbox.get()
            if stop:
                outbox.put((None,) * (out_width - 1) + (stop,))
                return
            outbox.put(tuple(stage(*fields)) + (stop,))

    def extract_skeletons(self):
        self._pump(self.infer_queue, self.track_queue, 10, self._estimate)

    def track(self):
        self._pump(self.track_queue, self.write_queue, 7, self._associate)

    def writeout(self):
        self._pump(self.write_queue, self.final_queue, 5, self._store)

    # stage 1: (color, depth, skeleton folder, intrinsics) ->
    #          (image, boxes, scores, keypoints, heatmaps, filtered_skel, prep s, infer s, color)
    def _estimate(self, color, depth, skeleton_folder, intr_mat):
        times = extract_2dskeletons(args=self.args, pose_extractor=self.PE, color_src=color, depth_src=depth,
                                    intr_mat=intr_mat, skeleton_folder=skeleton_folder,
                                    enable_timer=self.enable_timer)
        pyop = self.PE.pyop
        # copies: the estimator overwrites its buffers with the next frame while the tracker thread reads these
        found = [None] * 5 if pyop.pose_empty else \
            [np.copy(pyop.pose_bounding_box), np.copy(pyop.pose_scores), np.copy(pyop.pose_keypoints),
             np.copy(pyop.pose_heatmaps), pyop.filtered_skel]
        return (np.copy(pyop.opencv_image), *found, times['data_prep_time'] + 1e-8, times['infer_time'] + 1e-8, color)

    # stage 2: -> (image, filtered_skel, prep s, infer s, track s | -1, color)
    def _associate(self, image, boxes, scores, keypoints, heatmaps, filtered_skel, prep_s, infer_s, color):
        annotate = self.args.op_save_track_image
        if boxes is None or self.TK is None:  # nobody in the frame (or no tracker): the tracks only age
            if self.TK is not None:
                self.TK.no_measurement_predict_and_update()
            track_s = -1
            if annotate:
                image = PYOP._draw_bounding_box_on_skeleton_image(PYOP._draw_text_on_skeleton_image(image, scores), boxes)
        else:
            with Timer("update", self.enable_timer, False) as clock:
                self.TK.update(boxes, scores, keypoints, heatmaps, (self.args.op_image_width, self.args.op_image_height))
            track_s = clock.duration + 1e-8
            if annotate:
                image = PYOP._draw_tracking_bounding_box_image(PYOP._draw_text_on_skeleton_image(image, scores),
                                                               self.TK.tracks)
        return image, filtered_skel, prep_s, infer_s, track_s, color

    # stage 3: -> (filtered_skel, prep s, infer s, track s); writes <root>/<sensor>_<tracker name>/<file>
    def _store(self, image, filtered_skel, prep_s, infer_s, track_s, color):
        if self.args.op_save_track_image and filtered_skel is not None:
            sensor = color.split('/')[-2]
            target = color.replace(sensor, f"{sensor}_{getattr(self.TK, 'name', 'notracker')}")
            os.makedirs(os.path.dirname(target), exist_ok=True)
            _write_image(target, image)
        return filtered_skel, prep_s, infer_s, track_s


def _sorted_files(base_path: str) -> List[str]:
    assert os.path.isdir(base_path), f'{base_path} does not exist...'
    return sorted(os.path.join(base_path, i) for i in os.listdir(base_path))


def _skeleton_folder(color_filepath: str) -> str:
    # <root>/<sensor>/<file> -> <root>/skeleton
    return os.path.join(os.path.split(os.path.split(color_filepath)[0])[0], 'skeleton')


def extract_2dskeletons_online(args: argparse.Namespace, pose_extractor: Optional[OpenPosePoseExtractor] = None,
                               max_passes: Optional[int] = None) -> None:
    """Walks the folder again and again (new files appear while a camera records) until the
    display reports a stop (``inference.py:285-352``).  ``max_passes`` (addition) bounds the
    number of walks; the reference loops until its window is closed."""
    base_path = args.op_input_color_image
    assert os.path.isdir(base_path), f'{base_path} does not exist...'
    PE = pose_extractor or OpenPosePoseExtractor(args)
    count, passes, break_loop = 0, 0, False
    while not break_loop:
        for color_filepath in _sorted_files(base_path):
            count += 1
            infer_out = extract_2dskeletons(args=args, pose_extractor=PE, color_src=color_filepath,
                                            skeleton_folder=_skeleton_folder(color_filepath), enable_timer=True)
            status = PE.display(win_name='000', speed=1, scale=args.op_display)
            if not status[0]:
                break_loop = True
            if count % 100 == 0:
                print(f"Image : {color_filepath.split('/')[-1]} | "
                      f"#Skel filtered : {PE.pyop.filtered_skel} | "
                      f"Prep time : {infer_out['data_prep_time'] + 1e-8:.3f} | "
                      f"Pose time : {infer_out['infer_time'] + 1e-8:.3f}")
            if break_loop:
                break
        passes += 1
        if max_passes is not None and passes >= max_passes:
            break


def extract_2dskeletons_offline(args: argparse.Namespace,
                                pose_extractor: Optional[OpenPosePoseExtractor] = None) -> dict:
    """One pass over the folder, frame by frame, with a progress line (``inference.py:355-427``).
    The frame-rate average restarts at frame 15 (warm-up).  Returns the runtime dict
    ``{'PE': [1/infer_time...], 'TK': []}`` (addition: the reference returns nothing)."""
    filepaths = _sorted_files(args.op_input_color_image)
    runtime = {'PE': [], 'TK': []}
    PE = pose_extractor or OpenPosePoseExtractor(args)
    bar = _progress(filepaths)
    for idx, color_filepath in enumerate(bar):
        break_loop = idx + 1 == len(filepaths)
        if idx == 15:
            runtime = {'PE': [], 'TK': []}
        infer_out = extract_2dskeletons(args=args, pose_extractor=PE, color_src=color_filepath,
                                        skeleton_folder=_skeleton_folder(color_filepath), enable_timer=True)
        status = PE.display(win_name='000', speed=1, scale=args.op_display)
        if not status[0]:
            break_loop = True
        infer_time = infer_out['infer_time'] + 1e-8
        prep_time = infer_out['data_prep_time'] + 1e-8
        runtime['PE'].append(1 / infer_time)
        bar.set_description(f"Image : {color_filepath.split('/')[-1]} | "
                            f"#Skel filtered : {PE.pyop.filtered_skel} | "
                            f"Prep time : {prep_time:.3f} | "
                            f"Pose time : {infer_time:.3f} | "
                            f"FPS PE : {sum(runtime['PE']) / len(runtime['PE']):.3f}")
        if break_loop:
            break
    return runtime


def extract_2dskeletons_offline_batched(args: argparse.Namespace,
                                        pose_extractor: Optional[OpenPosePoseExtractor] = None,
                                        decode_workers: int = 8) -> dict:
    """Addition: the offline folder walk with ``args.posehip_batch`` frames per engine call.
    Files are decoded by a small thread pool one batch ahead; every frame then goes through
    the same filter and writers as :func:`extract_2dskeletons` (same CSV / image files).
    Returns ``{'frames', 'failed', 'seconds', 'filtered_skel': [...]}``."""
    import time
    filepaths = _sorted_files(args.op_input_color_image)
    batch = max(1, int(getattr(args, 'posehip_batch', None) or 1))
    PE = pose_extractor or OpenPosePoseExtractor(args)
    pyop = PE.pyop
    h, w = args.op_image_height, args.op_image_width

    def load(path):
        try:
            return read_color_file(path).reshape(h, w, 3)
        except Exception as e:
            print(e)
            print(f"[WARN] : Error loading data, skipping {path}")
            return None

    out = {'frames': 0, 'failed': 0, 'seconds': 0.0, 'filtered_skel': []}
    chunks = [filepaths[i:i + batch] for i in range(0, len(filepaths), batch)]
    t0 = time.time()
    with ThreadPoolExecutor(max_workers=decode_workers) as pool:
        pending = [pool.submit(load, p) for p in chunks[0]] if chunks else []
        for ci, chunk in enumerate(chunks):
            images = [f.result() for f in pending]
            pending = [pool.submit(load, p) for p in chunks[ci + 1]] if ci + 1 < len(chunks) else []
            good = [(p, im) for p, im in zip(chunk, images) if im is not None]
            out['failed'] += len(chunk) - len(good)
            if not good:
                continue
            pyop.predict_batch([im for _, im in good])
            for (path, _), datum in zip(good, pyop.last_datums):
                pyop.adopt(datum)
                pyop.filter_prediction()
                prefix = ''
                paths = prepare_save_paths(_skeleton_folder(path), prefix, args.op_save_skel,
                                           args.op_save_skel_image, args.op_save_3d_skel)
                if paths['skel_save_path'] is not None:
                    pyop.save_pose_keypoints(paths['skel_save_path'])
                if paths['skel_image_save_path'] is not None:
                    pyop.save_skeleton_image(paths['skel_image_save_path'], None)
                out['filtered_skel'].append(pyop.filtered_skel)
                out['frames'] += 1
    out['seconds'] = time.time() - t0
    return out


def extract_2dskeletons_and_track_offline(args: argparse.Namespace, tracker=None,
                                          pose_extractor: Optional[OpenPosePoseExtractor] = None) -> dict:
    """Folder walk through :class:`ExtractSkeletonAndTrack` (``inference.py:430-534``).  Returns the
    runtime dict (addition)."""
    filepaths = _sorted_files(args.op_input_color_image)
    delay_switch, delay_counter = 0, 0  # delay = predict without update in the tracker
    runtime = {'PE': [], 'TK': []}
    PE = pose_extractor or OpenPosePoseExtractor(args)
    if tracker is None:
        try:
            from tracking.track import Tracker  # the caller's own tree (out of scope here)
        except ImportError as e:
            raise ImportError("no tracker given and tracking.track.Tracker is not importable; pass "
                              "tracker=<object with update/no_measurement_predict_and_update/tracks/name>") from e
        tracker = Tracker(args, 30 // (delay_switch + 1))
    EST = ExtractSkeletonAndTrack(args, PE, tracker, True)
    EST.start()
    bar = _progress(filepaths)
    for idx, color_filepath in enumerate(bar):
        break_loop = idx + 1 == len(filepaths)
        if idx == 15:
            runtime = {'PE': [], 'TK': []}
        if delay_counter > 0:
            delay_counter -= 1
            _image = read_color_file(color_filepath)
            EST.TK.no_measurement_predict_and_update()
            EST.PE.display(win_name='000', speed=1, scale=args.op_display, image=_image, bounding_box=True,
                           tracks=EST.TK.tracker.tracks)
        else:
            delay_counter = delay_switch
            EST.queue_input((color_filepath, None, _skeleton_folder(color_filepath), None, False))
            kept_total, prep_time, infer_time, track_time, _ = EST.queue_output()
            status = EST.PE.display(win_name='000', speed=1, scale=args.op_display, image=None,
                                    bounding_box=False, tracks=EST.TK.tracks)
            if not status[0]:
                break_loop = True
            runtime['PE'].append(1 / infer_time)
            runtime['TK'].append(1 / track_time)
            bar.set_description(f"Image : {color_filepath.split('/')[-1]} | "
                                f"#Skel filtered : {kept_total} | "
                                f"#Tracks : {len(EST.TK.tracks)} | "
                                f"Prep time : {prep_time:.3f} | "
                                f"Pose time : {infer_time:.3f} | "
                                f"Track time : {track_time:.3f} | "
                                f"FPS PE : {sum(runtime['PE']) / len(runtime['PE']):.3f} | "
                                f"FPS TK : {sum(runtime['TK']) / len(runtime['TK']):.3f}")
        if break_loop:
            EST.break_process_loops()
            break
    return runtime


def compare_tracker(arg_op: argparse.Namespace, tracker=None) -> None:
    """``inference.py:537-574``: the ByteTrack run on the reference's MOT17 sample folder."""
    arg_op.op_input_color_image = "data/mot17/dev1/100/color"
    arg_op.op_save_track_image = False
    print(">>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>>> op_track_bytetrack")
    arg_op.op_track_bytetrack = True
    extract_2dskeletons_and_track_offline(arg_op, tracker=tracker)
    arg_op.op_track_bytetrack = False


def main(argv=None) -> None:
    """Mode dispatch of ``inference.py:577-601``."""
    arg_op, _ = get_parser().parse_known_args(argv)
    if (arg_op.op_track_deepsort or arg_op.op_track_bytetrack or arg_op.op_track_ocsort or
            arg_op.op_track_strongsort):
        extract_2dskeletons_and_track_offline(arg_op)
    elif arg_op.op_mode == 'single':
        PE = OpenPosePoseExtractor(arg_op)
        extract_2dskeletons(args=arg_op, pose_extractor=PE, color_src=arg_op.op_input_color_image,
                            skeleton_folder=arg_op.op_save_skel_folder, enable_timer=True)
    elif arg_op.op_mode == 'online':
        extract_2dskeletons_online(arg_op)
    elif arg_op.op_mode == 'offline':
        if (getattr(arg_op, 'posehip_batch', None) or 1) > 1:
            extract_2dskeletons_offline_batched(arg_op)
        else:
            extract_2dskeletons_offline(arg_op)
    elif arg_op.op_mode == 'comparetracker':
        compare_tracker(arg_op)
    else:
        raise ValueError("Unknown op_mode")
    print("[INFO] : FINISHED")


if __name__ == "__main__":
    main()
