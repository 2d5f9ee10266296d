"""replaceBN (pc_processor/layers/sync_bn.py:326-368).

Under DistributedDataParallel the reference's _SynchronizedBatchNorm never synchronises: its parallel path is
enabled only by the nn.DataParallel replicate callback (sync_bn.py:53,86-94), so every rank normalises with its
LOCAL batch statistics.  The HIP plan implements exactly that (per-rank statistics from the conv
epilogue), therefore replaceBN is the identity here; it exists so tasks/pmf/trainer.py:36-37 runs unmodified."""


def replaceBN(model):
    return model
