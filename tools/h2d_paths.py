"""Which way do 64 pageable 10 s utterances reach the device fastest?
python tools/h2d_paths.py"""
import concurrent.futures
import time

import numpy as np
import torch

COUNT, SAMPLES = 64, 160000
pool = concurrent.futures.ThreadPoolExecutor(8)


def measure(name, fn, rounds=10):
    for _ in range(2):
        fn()
    torch.cuda.synchronize()
    start = time.perf_counter()
    for _ in range(rounds):
        fn()
    torch.cuda.synchronize()
    ms = (time.perf_counter() - start) / rounds * 1e3
    print(f'{name:58s} {ms:7.2f} ms')


for dtype in (torch.float32, torch.int16):
    item = torch.tensor([], dtype=dtype).element_size()
    audios = [torch.randn(SAMPLES).to(dtype) for _ in range(COUNT)]
    device = torch.empty(COUNT * SAMPLES, dtype=dtype, device='cuda')
    pinned = torch.empty(COUNT * SAMPLES, dtype=dtype).pin_memory()
    pageable = torch.empty(COUNT * SAMPLES, dtype=dtype)
    stream = torch.cuda.Stream()
    print(f'--- {dtype}, {COUNT * SAMPLES * item / 1e6:.1f} MB')

    def sequential():
        for i, a in enumerate(audios):
            device[i * SAMPLES:(i + 1) * SAMPLES].copy_(a, non_blocking=True)
    measure('sequential copy_ per utterance, pageable -> device', sequential)

    def threaded_direct():
        def work(k):
            with torch.cuda.stream(stream):
                for i in range(k, COUNT, 8):
                    device[i * SAMPLES:(i + 1) * SAMPLES].copy_(
                        audios[i], non_blocking=True)
        list(pool.map(work, range(8)))
    measure('8 threads copy_ per utterance, pageable -> device', threaded_direct)

    def gather_pinned():
        def work(k):
            for i in range(k * 8, k * 8 + 8):
                pinned[i * SAMPLES:(i + 1) * SAMPLES].copy_(audios[i])
        list(pool.map(work, range(8)))
        device.copy_(pinned, non_blocking=True)
    measure('8 threads gather into pinned + one DMA', gather_pinned)

    def gather_pinned_numpy():
        view = pinned.numpy()
        def work(k):
            for i in range(k * 8, k * 8 + 8):
                np.copyto(view[i * SAMPLES:(i + 1) * SAMPLES], audios[i].numpy())
        list(pool.map(work, range(8)))
        device.copy_(pinned, non_blocking=True)
    measure('8 threads numpy gather into pinned + one DMA', gather_pinned_numpy)

    def gather_pageable():
        def work(k):
            for i in range(k * 8, k * 8 + 8):
                pageable[i * SAMPLES:(i + 1) * SAMPLES].copy_(audios[i])
        list(pool.map(work, range(8)))
        device.copy_(pageable, non_blocking=True)
    measure('8 threads gather into pageable + one copy_', gather_pageable)

    def dma_only():
        device.copy_(pinned, non_blocking=True)
    measure('one DMA from pinned (no gather)', dma_only)

    def cat_copy():
        device.copy_(torch.cat(audios), non_blocking=True)
    measure('torch.cat + one copy_', cat_copy)

    def single_gather_pinned():
        for i, a in enumerate(audios):
            pinned[i * SAMPLES:(i + 1) * SAMPLES].copy_(a)
    measure('1 thread gather into pinned (no DMA)', single_gather_pinned)
