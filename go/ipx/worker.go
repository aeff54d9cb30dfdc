package ipx

/*
#include <stdlib.h>
#include "ipx.h"
*/
import "C"

import (
	"unsafe"
)

// Batcher is the micro-batcher of the library (ipx_batcher_*, include/ipx.h): the goroutines of internal/worker/worker.go:112-149 pull ONE
// message each from a channel of concurrency*2 (:88); every one of them hands its file over with Process and blocks until ITS objects are
// back.  Grouping by frame size and operator content, the flush rules (size, timer, at once while the pool has a free feeder) and the
// per-file status live below the ABI (and are tested
// there: tests/test_batcher_gpu.py, tools/sanitize/batcher_host_test.cpp under ThreadSanitizer) -- round 2's Go-side batching is gone.
// At-least-once semantics are unchanged: processMessage (worker.go:165-234) commits its message only after Process returned and
// fileRepo.SaveProcessed stored the objects.
type Batcher struct {
	c *C.ipx_batcher
	p *Pool
}

// NewBatcher: maxBatch files per job (0 = 256), maxWaitMicros how long the first file of a group waits for company while the pool is
// busy (0 = 2000; with WORKER_CONCURRENCY = 3 goroutines a file leaves at once: 2.4 ms p50 per message on one MI355X),
// a group = files of one frame size, one JPEG shape (components, luma sampling) and one operator content;
// quality = domain.DefaultJPEGQuality (task.go:57; 0 = 85).
func NewBatcher(p *Pool, maxBatch, maxWaitMicros, quality int) (*Batcher, error) {
	cfg := C.ipx_batcher_config{max_batch: C.int32_t(maxBatch), max_wait_us: C.int32_t(maxWaitMicros), quality: C.int32_t(quality)}
	var b *C.ipx_batcher
	if err := call(func() C.int { return C.ipx_batcher_create(p.c, &cfg, &b) }); err != nil {
		return nil, err
	}
	return &Batcher{b, p}, nil
}

// Close flushes what is pending and waits for it.
func (b *Batcher) Close() { C.ipx_batcher_destroy(b.c); b.c = nil }

// Objects are the three streams of one message (nil for operators its task did not ask for).  They are views into blocks the library
// owns: copy or store them (fileRepo.SaveProcessed, image_processor.go:76), then call Release.
type Objects struct {
	Resize, Thumbnail, Watermark []byte
	release                      func()
}

func (o *Objects) Release() {
	if o.release != nil {
		o.release()
		o.release = nil
	}
}

// Process is the drop-in for (*ImageProcessor).Process on the JPEG path (image_processor.go:39-102): the object bytes GetOriginal
// returned (worker.go:177-186), the frame size from the file's header (image.DecodeConfig) and the operators of the task.  It blocks
// until the file's batch has run.  IsUnsupported(err): this file is not one the GPU path decodes (progressive CMYK, 4:1:1, ...) -- run the
// reference's own image.Decode path for this message; its neighbours in the batch are not affected.
func (b *Batcher) Process(file []byte, w, h int, o Ops) (*Objects, error) {
	ops, free := b.p.ops(w, h, o)
	defer free() // copied by ipx_batcher_submit
	cfile := C.CBytes(file)
	defer C.free(cfile) // read until ipx_batcher_wait has returned
	fb := C.ipx_bytes{data: (*C.uint8_t)(cfile), len: C.size_t(len(file))}
	var t C.ipx_batch_ticket
	if err := call(func() C.int { return C.ipx_batcher_submit(b.c, &fb, &ops, &t) }); err != nil {
		return nil, err
	}
	var res C.ipx_batch_result
	if err := call(func() C.int { return C.ipx_batcher_wait(b.c, t, &res) }); err != nil {
		C.ipx_batcher_release(b.c, t)
		return nil, err
	}
	if Status(res.status) != OK {
		C.ipx_batcher_release(b.c, t)
		return nil, &Error{Status(res.status), "file not decodable on the GPU path"}
	}
	return &Objects{Resize: view(res.resize), Thumbnail: view(res.thumb), Watermark: view(res.wm),
		release: func() { C.ipx_batcher_release(b.c, t) }}, nil
}

// Stats: how the files were grouped so far.
type BatcherStats struct{ Files, Batches, BySize, ByTimer, Largest, Pending, WhenIdle int64 }

func (b *Batcher) Stats() BatcherStats {
	var s C.ipx_batcher_stats
	C.ipx_batcher_get_stats(b.c, &s)
	return BatcherStats{int64(s.files), int64(s.batches), int64(s.flushed_by_size), int64(s.flushed_by_timer), int64(s.largest_batch), int64(s.pending_files), int64(s.flushed_when_idle)}
}

var _ = unsafe.Pointer(nil)
