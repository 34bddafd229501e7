"""MI355X-native isolated-word recognition engine: the hot path of gk969/stm32-speech-recognition
(noise_atap -> VAD -> get_mfcc -> dtw -> spch_recg) as hand-written gfx950 kernels behind a C ABI
(include/sr_engine.h).  `engine` = batched API, `compat` = the reference's scalar entry points,
`synth` = synthetic capture buffers for tests and bench."""
from .engine import Engine, SrError, load_library, DIS_ERR, ST_OK, ST_VAD_FAIL, ST_MFCC_FAIL, ST_SEG_OOB  # noqa: F401
