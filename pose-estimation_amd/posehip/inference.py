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

    with Timer("data preparation", enable_timer, printout=False) as t1:
        paths = prepare_save_paths(skeleton_folder, skeleton_file_prefix, args.op_save_skel,
                                   args.op_save_skel_image, args.op_save_3d_skel)
        if isinstance(color_src, str):
            try:
                image = read_color_file(color_src)
            except Exception as e:
                print(e)
                print(f"[WARN] : Error loading data, skipping {color_src}")
                return dict(_FAILED)
        else:
            assert isinstance(color_src, np.ndarray)
            image = color_src

        depth = None
        if args.op_extract_3d_skel and depth_src is not None:
            if isinstance(depth_src, str):
                try:
                    depth = read_depth_file(depth_src)
                except Exception as e:
                    print(e)
                    print(f"[WARN] : Error loading data, skipping {depth_src}")
                    return dict(_FAILED)
            else:
                assert isinstance(depth_src, np.ndarray)
                depth = depth_src

        try:
            image = image.reshape(args.op_image_height, args.op_image_width, 3)
        except Exception as e:
            print(e)
            print("[WARN] : Error in reshaping image...")
            if isinstance(color_src, str):
                print(f"skipping {color_src}")
            return dict(_FAILED)

    with Timer("infer", enable_timer, printout=False) as t2:
        try:
            if args.op_extract_3d_skel:
                pose_extractor.predict_3d(image, depth, intr_mat, paths['skel_save_path'],
                                          paths['3dskel_save_path'], paths['skel_image_save_path'])
            else:
                pose_extractor.predict(image, depth, paths['skel_save_path'], paths['skel_image_save_path'])
        except Exception as e:
            print(e)
            print("[WARN] : Error in pose extraction...")
            if isinstance(color_src, str):
                print(f"skipping {color_src}")
            return dict(_FAILED)

    return {'status': True, 'data_prep_time': t1.duration, 'infer_time': t2.duration}


class ExtractSkeletonAndTrack:
    """Estimator -> tracker -> writer as three threads joined by queues of depth 10
    (``inference.py:22-165``).

    queue_input((color_path, depth_path, skeleton_folder, intr_mat, break_loop))
    queue_output() -> (filtered_skel, prep_time, infer_time, track_time, break_loop)

    ``break_process_loops()`` sends the sentinel through all three stages and waits for it.
    """

    def __init__(self, args, pose_extractor, tracker, enable_timer) -> None:
        self.args = args
        self.PE = pose_extractor
        self.TK = tracker
        self.enable_timer = enable_timer
        if args.op_proc == 'mp':
            raise ValueError("mp not supported for now, openpose inference seems to cause the mp to hang.")
        self.infer_queue = Queue(maxsize=10)
        self.track_queue = Queue(maxsize=10)
        self.write_queue = Queue(maxsize=10)
        self.final_queue = Queue(maxsize=10)

    def start(self):
        if self.args.op_proc == 'mp':
            raise ValueError("mp not supported for now, openpose inference seems to cause the mp to hang.")
        self.p_extract = Thread(target=self.extract_skeletons, args=())
        self.p_track = Thread(target=self.track, args=())
        self.p_write = Thread(target=self.writeout, args=())
        for t in (self.p_extract, self.p_track, self.p_write):
            t.start()

    def queue_input(self, args):
        self.infer_queue.put(args)

    def queue_output(self):
        return self.final_queue.get()

    def break_process_loops(self):
        self.queue_input((None, None, None, None, True))
        self.queue_output()

    # stage 1: estimator ------------------------------------------------------------
    def extract_skeletons(self):
        while True:
            color_filepath, depth_filepath, skel_filepath, intr_mat, break_loop = self.infer_queue.get()
            if break_loop:
                self.track_queue.put((None,) * 9 + (break_loop,))
                break
            infer_out = extract_2dskeletons(args=self.args, pose_extractor=self.PE, color_src=color_filepath,
                                            depth_src=depth_filepath, intr_mat=intr_mat,
                                            skeleton_folder=skel_filepath, enable_timer=self.enable_timer)
            pyop = self.PE.pyop
            if pyop.pose_empty:
                boxes = scores = keypoints = heatmaps = filered_skel = None
            else:  # copies: the estimator's buffers are overwritten by the next frame
                boxes = pyop.pose_bounding_box.copy()
                scores = pyop.pose_scores.copy()
                keypoints = pyop.pose_keypoints.copy()
                heatmaps = pyop.pose_heatmaps.copy()
                filered_skel = pyop.filtered_skel
            img = pyop.opencv_image.copy()
            self.track_queue.put((img, boxes, scores, keypoints, heatmaps, filered_skel,
                                  infer_out['data_prep_time'] + 1e-8, infer_out['infer_time'] + 1e-8,
                                  color_filepath, break_loop))

    # stage 2: tracker ---------------------------------------------------------------
    def track(self):
        while True:
            (img, boxes, scores, keypoints, heatmaps, filered_skel, prep_time, infer_time, color_fp,
             break_loop) = self.track_queue.get()
            if break_loop:
                self.write_queue.put((None,) * 6 + (break_loop,))
                break
            if boxes is not None and self.TK is not None:
                with Timer("update", self.enable_timer, False) as t:
                    image_size = (self.args.op_image_width, self.args.op_image_height)
                    self.TK.update(boxes, scores, keypoints, heatmaps, image_size)
                if self.args.op_save_track_image:
                    img = PYOP._draw_text_on_skeleton_image(img, scores)
                    img = PYOP._draw_tracking_bounding_box_image(img, self.TK.tracks)
                duration = t.duration + 1e-8
            else:
                if self.TK is not None:
                    self.TK.no_measurement_predict_and_update()
                if self.args.op_save_track_image:
                    img = PYOP._draw_text_on_skeleton_image(img, scores)
                    img = PYOP._draw_bounding_box_on_skeleton_image(img, boxes)
                duration = -1
            self.write_queue.put((img, filered_skel, prep_time, infer_time, duration, color_fp, break_loop))

    # stage 3: writer ------------------------------------------------------------------
    def writeout(self):
        while True:
            img, filered_skel, prep_time, infer_time, track_time, color_fp, break_loop = self.write_queue.get()
            if break_loop:
                self.final_queue.put((None, None, None, None, break_loop))
                break
            if self.args.op_save_track_image and filered_skel is not None:
                parts = color_fp.split('/')
                name = getattr(self.TK, 'name', 'notracker')
                save_file = color_fp.replace(parts[-2], f'{parts[-2]}_{name}')
                os.makedirs(os.path.dirname(save_file), exist_ok=True)
                _write_image(save_file, img)
            self.final_queue.put((filered_skel, prep_time, infer_time, track_time, break_loop))


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
            filered_skel, prep_time, infer_time, track_time, _ = EST.queue_output()
            status = EST.PE.display(win_name='000', speed=1, scale=args.op_display, image=None,
                                    bounding_box=False, tracks=EST.TK.tracks)
            if not status[0]:
                break_loop = True
            runtime['PE'].append(1 / infer_time)
            runtime['TK'].append(1 / track_time)
            bar.set_description(f"Image : {color_filepath.split('/')[-1]} | "
                                f"#Skel filtered : {filered_skel} | "
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
