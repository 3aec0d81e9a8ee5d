"""Empty stand-in for the `models` package that AVS/run_adapt_avs.py:14, AVQA/run_adapt_avqa.py:18 and AVS/test.py:4 import and never
use (a leftover of the CAV-MAE code base the runners were derived from; the reference tree ships no such package, so those
runners cannot start without one).  Nothing of the hot path lives here."""
