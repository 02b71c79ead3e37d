"""masa-cudalign_amd: MI355X-native Stage-1 SW/NW strip-wavefront engine (drop-in for the
MASA-CUDAlign Stage-1 hot path).  The product is csrc/ (HIP kernels + C ABI, built in-tree as
libmi355sw.so); this package is the thin Python host mirror used by tests/ and bench.py.

The directory name contains a hyphen, import it through `load_package()` of the repo-root
`__graft_entry__.py` (registers it as `masa_cudalign_amd`).
"""
from .engine import (  # noqa: F401
    MI355Aligner, AlignerError, Partition, StreamParams, LIB_PATH, load_library, build_library,
    INF, NEEDLEMAN_WUNSCH, SMITH_WATERMAN,
    INIT_WITH_ZEROES, INIT_WITH_GAPS, INIT_WITH_CUSTOM_DATA, INIT_WITH_GAPS_OPENED,
)
from .manager import (  # noqa: F401
    Stage1Manager, BestScoreList, InitialCellsReader,
    AT_NOWHERE, AT_ANYWHERE, AT_SEQUENCE_1, AT_SEQUENCE_2, AT_SEQUENCE_1_OR_2, AT_SEQUENCE_1_AND_2,
)
from . import seqgen  # noqa: F401
from . import sra  # noqa: F401
from .stage1 import stage1  # noqa: F401
from .manager import AlignerManager, BacktraceLost  # noqa: F401
from .stage2 import stage2  # noqa: F401
from .stage3 import stage3  # noqa: F401
from . import crosspoints, pipeline  # noqa: F401
