package ipx

/*
#include <stdlib.h>
#include "ipx.h"
*/
import "C"

import (
	"unsafe"
)

// Pool is one process over several GPUs (ipx_pool_*): one context per listed device, feeder threads with a stream each, ONE queue of
// chunks ordered by cost, largest first.  The reference's worker is exactly this shape -- one process, WORKER_CONCURRENCY goroutines
// pulling independent messages (worker.go:88-96) -- so a pool replaces the per-process "one GPU" limit without touching Kafka.
type Pool struct{ c *C.ipx_pool }

// NewPool lists the devices to use (e.g. 0..7 on an 8-GPU node); lanesPerDevice 0 = 3.
func NewPool(devices []int, lanesPerDevice int) (*Pool, error) {
	d := make([]C.int, len(devices))
	for i, v := range devices {
		d[i] = C.int(v)
	}
	cfg := C.ipx_pool_config{lanes_per_device: C.int32_t(lanesPerDevice)}
	var p *C.ipx_pool
	if err := call(func() C.int { return C.ipx_pool_create(&d[0], C.int(len(d)), &cfg, &p) }); err != nil {
		return nil, err
	}
	return &Pool{p}, nil
}

func (p *Pool) Close() { C.ipx_pool_destroy(p.c); p.c = nil }

// Pinned allocates staging on the NUMA node next to slot's GPU (the allocation and its first touch happen on a thread bound there).
func (p *Pool) Pinned(slot, n int) (*Pinned, error) {
	m := C.ipx_pool_host_alloc(p.c, C.int(slot), C.size_t(n))
	if m == nil {
		return nil, &Error{NoMem, C.GoString(C.ipx_last_error())}
	}
	return &Pinned{unsafe.Slice((*byte)(m), n), func() { C.ipx_pool_host_free(p.c, C.int(slot), m) }}, nil
}

// Job is a submitted batch; the goroutine that submitted it is free until it calls Wait.
type Job struct {
	p       *Pool
	ticket  C.ipx_ticket
	n       int
	cmem    []unsafe.Pointer // C copies that have to live until Release
	resize  []C.ipx_bytes
	thumb   []C.ipx_bytes
	wm      []C.ipx_bytes
	cstatus *[1 << 24]C.int32_t
}

func (p *Pool) ops(w, h int, o Ops) (C.ipx_pool_ops, func()) {
	var c C.ipx_pool_ops
	c.sw, c.sh = C.int32_t(w), C.int32_t(h)
	if o.Resize != nil {
		c.do_resize, c.resize_w, c.resize_h, c.keep_aspect = 1, C.int32_t(o.Resize.W), C.int32_t(o.Resize.H), b2i(o.Resize.KeepAspect)
	}
	if o.Thumb != nil {
		c.do_thumbnail, c.thumb_size, c.crop_to_fit = 1, C.int32_t(o.Thumb.Size), b2i(o.Thumb.CropToFit)
	}
	free := func() {}
	if o.Watermark {
		c.do_watermark = 1
		c.glyphs, free = cGlyphs(o.Glyphs) // copied by ipx_job_submit: freed right after it returns
		c.n_glyphs = C.int32_t(len(o.Glyphs))
		for i := 0; i < 4; i++ {
			c.col[i] = C.uint8_t(o.Color[i])
		}
	}
	return c, free
}

// PixelKind names the packed image type of a pixel job's frames (Pix as Go holds it).
type PixelKind int

const (
	PixRGBA    PixelKind = C.IPX_JOB_RGBA8
	PixNRGBA   PixelKind = C.IPX_JOB_NRGBA8
	PixGray    PixelKind = C.IPX_JOB_GRAY8
	PixNRGBA64 PixelKind = C.IPX_JOB_NRGBA64
	PixRGBA64  PixelKind = C.IPX_JOB_RGBA64
	PixGray16  PixelKind = C.IPX_JOB_GRAY16
	PixCMYK    PixelKind = C.IPX_JOB_CMYK
)

func (k PixelKind) bytesPerPixel() int {
	switch k {
	case PixGray:
		return 1
	case PixGray16:
		return 2
	case PixNRGBA64, PixRGBA64:
		return 8
	}
	return 4
}

// SubmitPixels queues n decoded RGBA8 frames of w x h.  The job is ASYNCHRONOUS: feeder threads read src and write the outputs after
// this call has returned, so all four are *Pinned (Pool.Pinned: C memory the garbage collector neither moves nor frees; nil = that
// operator's output is not wanted) and must stay allocated until Wait has returned.  A Go slice here would be read after the call
// that named it -- cgo cannot check that, hence the type.
func (p *Pool) SubmitPixels(w, h, n int, o Ops, src, resizeOut, thumbOut, wmOut *Pinned, resizeBytes, thumbBytes int) (*Job, error) {
	return p.SubmitPixelsOf(PixRGBA, w, h, n, o, src, resizeOut, thumbOut, wmOut, resizeBytes, thumbBytes)
}

func pinnedPtr(m *Pinned) *C.uint8_t {
	if m == nil {
		return nil
	}
	return ptr(m.Bytes)
}

// SubmitPixelsOf does the same for frames of any packed type image.Decode returns (rows packed without stride padding).
func (p *Pool) SubmitPixelsOf(kind PixelKind, w, h, n int, o Ops, src, resizeOut, thumbOut, wmOut *Pinned, resizeBytes, thumbBytes int) (*Job, error) {
	ops, free := p.ops(w, h, o)
	defer free()
	bpp := kind.bytesPerPixel()
	j := C.ipx_job{kind: C.int32_t(kind), ops: ops, n: C.int32_t(n), src: pinnedPtr(src), sstride: C.int32_t(w * bpp), src_frame_stride: C.size_t(w * h * bpp),
		resize_out: pinnedPtr(resizeOut), resize_frame_stride: C.size_t(resizeBytes), thumb_out: pinnedPtr(thumbOut), thumb_frame_stride: C.size_t(thumbBytes),
		wm_out: pinnedPtr(wmOut), wm_frame_stride: C.size_t(w * h * 4)}
	job := &Job{p: p, n: n}
	if err := call(func() C.int { return C.ipx_job_submit(p.c, &j, &job.ticket) }); err != nil {
		return nil, err
	}
	return job, nil
}

// SubmitJPEG queues uploaded JPEG objects of w x h (worker.go:165-194: the bytes GetOriginal returned).  The files are copied into C memory
// here, so the caller's slices are free at once; outputs are views into blocks the pool owns until Release.
func (p *Pool) SubmitJPEG(w, h int, o Ops, files [][]byte, quality int) (*Job, error) {
	n := len(files)
	ops, free := p.ops(w, h, o)
	defer free()
	job := &Job{p: p, n: n}
	cf := (*[1 << 24]C.ipx_bytes)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(C.ipx_bytes{}))))
	job.cmem = append(job.cmem, unsafe.Pointer(cf))
	for i, f := range files {
		b := C.CBytes(f)
		job.cmem = append(job.cmem, b)
		cf[i] = C.ipx_bytes{data: (*C.uint8_t)(b), len: C.size_t(len(f))}
	}
	job.cstatus = (*[1 << 24]C.int32_t)(C.malloc(C.size_t(n) * 4))
	job.cmem = append(job.cmem, unsafe.Pointer(job.cstatus))
	// the output arrays are written by feeder threads after this call returns: they must be C memory too
	ro := (*C.ipx_bytes)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.ipx_bytes{}))))
	to := (*C.ipx_bytes)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.ipx_bytes{}))))
	wo := (*C.ipx_bytes)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.ipx_bytes{}))))
	job.cmem = append(job.cmem, unsafe.Pointer(ro), unsafe.Pointer(to), unsafe.Pointer(wo))
	job.resize, job.thumb, job.wm = unsafe.Slice(ro, n), unsafe.Slice(to, n), unsafe.Slice(wo, n)
	j := C.ipx_job{kind: C.IPX_JOB_JPEG, ops: ops, n: C.int32_t(n), files: &cf[0], quality: C.int32_t(quality),
		resize_jpeg: ro, thumb_jpeg: to, wm_jpeg: wo, status: &job.cstatus[0]}
	if !o.Watermark {
		j.wm_jpeg = nil
	}
	if o.Resize == nil {
		j.resize_jpeg = nil
	}
	if o.Thumb == nil {
		j.thumb_jpeg = nil
	}
	if err := call(func() C.int { return C.ipx_job_submit(p.c, &j, &job.ticket) }); err != nil {
		job.freeC()
		return nil, err
	}
	return job, nil
}

// Done never blocks.
func (j *Job) Done() bool {
	var d C.int
	C.ipx_job_poll(j.p.c, j.ticket, &d)
	return d != 0
}

// Wait blocks until every chunk of the job has run; the error is the first failing chunk's.
func (j *Job) Wait() error {
	return call(func() C.int { return C.ipx_job_wait(j.p.c, j.ticket, nil) })
}

// FileStatus (JPEG jobs, after Wait): OK, or why Go has to process file i itself.
func (j *Job) FileStatus(i int) Status { return Status(j.cstatus[i]) }
func (j *Job) Resize(i int) []byte     { return view(j.resize[i]) }
func (j *Job) Thumbnail(i int) []byte  { return view(j.thumb[i]) }
func (j *Job) Watermark(i int) []byte  { return view(j.wm[i]) }

func (j *Job) freeC() {
	for _, m := range j.cmem {
		C.free(m)
	}
	j.cmem = nil
}

// Release forgets the job and frees the blocks its JPEG outputs live in (call it after the objects have been saved).
func (j *Job) Release() {
	C.ipx_job_release(j.p.c, j.ticket)
	j.freeC()
}
