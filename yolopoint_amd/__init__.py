"""yolopoint_amd — the YOLOPoint hot path (backbone + heads forward -> box NMS / keypoint NMS ->
descriptor matching) as hand-written HIP kernels for MI355X (gfx950), behind the reference's own
Python module API.  See DESIGN.md / INTEGRATION.md.

Mirrors the reference import paths:
    models.Model / models.YOLOPoint            -> yolopoint_amd.models
    utils.utils.{flattenDetection,...}         -> yolopoint_amd.utils.utils
    utils.general_yolo.non_max_suppression     -> yolopoint_amd.utils.general_yolo
    evaluations.descriptor_evaluation.sample_desc_from_points -> yolopoint_amd.evaluations.descriptor_evaluation
    models.model_wrap.PointTracker             -> yolopoint_amd.models.model_wrap
"""
__version__ = "0.1.0"
