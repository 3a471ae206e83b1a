"""Device-side batch prefetching with the interface the train loops use (reference: BSRGAN/dataset.py:203-243,
``CUDAPrefetcher``: ``next()`` / ``reset()`` / ``len()``; batches are dicts of tensors, e.g. {"gt": ..., "lr": ...}).

The HIP kernels run on torch's current stream; the prefetcher stages the NEXT batch on its own copy stream and makes the
current stream wait for that copy before handing the batch over, so the host-to-device transfer of batch i+1 overlaps the
training iteration on batch i.  Tensors handed out are recorded on the consumer stream so the caching allocator does not
recycle them while kernels launched through the C ABI (which torch cannot see) may still read them.
"""
from __future__ import annotations

import torch


class CUDAPrefetcher:
    """``ingest_u8`` (an addition; the reference's loader hands out fp32 CHW tensors): a batch entry that is a uint8 (N, H, W, 3) tensor -- decoded
    images as cv2.imread returns them -- is copied as bytes and converted on the device on the copy stream (imgproc.image_to_tensor_u8:
    / 255, BGR -> RGB, HWC -> CHW), so the consumer sees the same fp32 (N, 3, H, W) batch at a quarter of the PCIe traffic."""

    def __init__(self, dataloader, device: torch.device, ingest_u8: bool = False, bgr: bool = True):
        self.original_dataloader = dataloader
        self.ingest_u8, self.bgr = ingest_u8, bgr
        self.device = torch.device(device)
        self.stream = torch.cuda.Stream(device=self.device)
        self.batch_data = None
        self.data = None
        self.reset()

    def _stage(self) -> None:
        try:
            batch = next(self.data)
        except StopIteration:
            self.batch_data = None
            return
        with torch.cuda.stream(self.stream):
            self.batch_data = {k: (v.to(self.device, non_blocking=True) if torch.is_tensor(v) else v) for k, v in batch.items()}
            if self.ingest_u8:
                from .imgproc import image_to_tensor_u8
                for k, v in self.batch_data.items():
                    if torch.is_tensor(v) and v.dtype == torch.uint8 and v.dim() == 4 and v.shape[-1] == 3:
                        self.batch_data[k] = image_to_tensor_u8(v, bgr=self.bgr)      # launched on the copy stream (A.stream_ptr() = current stream)

    def next(self):
        """the staged batch (or None at the end of the epoch); starts staging the following one"""
        consumer = torch.cuda.current_stream(self.device)
        consumer.wait_stream(self.stream)
        batch = self.batch_data
        if batch is not None:
            for v in batch.values():
                if torch.is_tensor(v) and v.is_cuda:
                    v.record_stream(consumer)
        self._stage()
        return batch

    def reset(self) -> None:
        self.data = iter(self.original_dataloader)
        self._stage()

    def __len__(self) -> int:
        return len(self.original_dataloader)
