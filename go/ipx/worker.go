package ipx

import (
	"context"
	"sync"
	"time"
)

// Task is what processMessage (internal/worker/worker.go:165-234) has in hand after fileRepo.GetOriginal: the object bytes, the frame size
// read from the JPEG header (image.DecodeConfig) and the operators of domain.ProcessingTask.
type Task struct {
	ID     string
	File   []byte
	W, H   int
	Ops    Ops
	Result chan TaskResult // receives exactly one value
}

// TaskResult carries the three objects (nil for operators the task did not ask for) or the reason the CPU path has to run.
type TaskResult struct {
	Resize, Thumbnail, Watermark []byte
	Err                          error // IsUnsupported(Err): run the reference's own image.Decode path for this message
	release                      func()
}

// Release returns the pinned blocks the objects live in; call it after fileRepo.SaveProcessed (image_processor.go:76).
func (r *TaskResult) Release() {
	if r.release != nil {
		r.release()
	}
}

// Batcher micro-batches the messages the worker's goroutines pull from Kafka (worker.go:112-149) by frame size and operator set and
// runs each batch as one JPEG job of the pool: image.Decode, every operator and jpeg.Encode on the GPUs, at-least-once semantics
// unchanged (a message is committed by its goroutine only after its TaskResult arrived and the objects were saved).
type Batcher struct {
	Pool     *Pool
	MaxBatch int           // files per job (256 is a good start: a part of ipx_plan_run_jpeg_jpeg)
	MaxWait  time.Duration // how long the first message of a batch may wait for company (a few milliseconds)
	Quality  int           // domain.DefaultJPEGQuality = 85 (task.go:57)

	mu      sync.Mutex
	pending map[batchKey][]*Task
	timers  map[batchKey]*time.Timer
}

type batchKey struct {
	w, h int
	ops  string // a fingerprint of the operator parameters and the rasterised text
}

// Submit hands one message over; the caller then waits on t.Result.
func (b *Batcher) Submit(ctx context.Context, t *Task, opsFingerprint string) {
	k := batchKey{t.W, t.H, opsFingerprint}
	b.mu.Lock()
	if b.pending == nil {
		b.pending, b.timers = map[batchKey][]*Task{}, map[batchKey]*time.Timer{}
	}
	b.pending[k] = append(b.pending[k], t)
	full := len(b.pending[k]) >= b.MaxBatch
	if len(b.pending[k]) == 1 && !full {
		b.timers[k] = time.AfterFunc(b.MaxWait, func() { b.flush(k) })
	}
	b.mu.Unlock()
	if full {
		b.flush(k)
	}
}

func (b *Batcher) flush(k batchKey) {
	b.mu.Lock()
	batch := b.pending[k]
	delete(b.pending, k)
	if t := b.timers[k]; t != nil {
		t.Stop()
		delete(b.timers, k)
	}
	b.mu.Unlock()
	if len(batch) == 0 {
		return
	}
	files := make([][]byte, len(batch))
	for i, t := range batch {
		files[i] = t.File
	}
	job, err := b.Pool.SubmitJPEG(k.w, k.h, batch[0].Ops, files, b.Quality)
	if err == nil {
		err = job.Wait()
	}
	if err != nil { // the whole batch failed (e.g. a frame size beyond the GPU path): every message takes the CPU path
		for _, t := range batch {
			t.Result <- TaskResult{Err: err}
		}
		if job != nil {
			job.Release()
		}
		return
	}
	var once sync.Once
	left := int32(len(batch))
	var lmu sync.Mutex
	release := func() { // the blocks are shared by the batch: free them when the last message has saved its objects
		lmu.Lock()
		left--
		last := left == 0
		lmu.Unlock()
		if last {
			once.Do(job.Release)
		}
	}
	for i, t := range batch {
		if st := job.FileStatus(i); st != OK {
			t.Result <- TaskResult{Err: &Error{st, "file not decodable on the GPU path"}, release: release}
			continue
		}
		t.Result <- TaskResult{Resize: job.Resize(i), Thumbnail: job.Thumbnail(i), Watermark: job.Watermark(i), release: release}
	}
}
