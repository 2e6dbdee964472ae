"""Host->device cost of one batch (the boundary hands over host tensors in train_epoch): image + float targets
versus image + uint8 label map (targets encoded on the GPU)."""
import time, torch
x = torch.randn(4, 3, 620, 620)
t = torch.randn(4, 8, 620, 620)
lab = torch.randint(0, 255, (4, 620, 620), dtype=torch.uint8)
for pin in (False, True):
    a, b, c = (v.pin_memory() if pin else v for v in (x, t, lab))
    for name, tensors in (("image + fp32 targets", (a, b)), ("image + uint8 labels", (a, c))):
        for _ in range(3):
            [v.cuda(non_blocking=True) for v in tensors]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20):
            [v.cuda(non_blocking=True) for v in tensors]
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        mb = sum(v.numel() * v.element_size() for v in tensors) / 2**20
        print("%-22s %s: %6.2f ms for %5.1f MB (%5.1f GB/s)" % (name, "pinned  " if pin else "pageable", dt * 1e3, mb, mb / 1024 / dt))
