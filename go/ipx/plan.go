package ipx

/*
#include <stdlib.h>
#include "ipx.h"
*/
import "C"

import (
	"image"
	"unsafe"
)

// Resize / Thumb are the operator parameters after the reference's parsing (resize.go:26-59, thumbnail.go:25-47).
type Resize struct {
	W, H       int
	KeepAspect bool
}
type Thumb struct {
	Size      int // 0 = domain.DefaultThumbnailSize (200, task.go:56)
	CropToFit bool
}

// Ops is what one task asks for (image_processor.go:104-127: every operator reads the ORIGINAL frame).
type Ops struct {
	Resize    *Resize
	Thumb     *Thumb
	Watermark bool
	Glyphs    []Glyph  // the rasterised text (may be empty: copy only)
	Color     [4]uint8 // parseColor's color.RGBA, not premultiplied (watermark.go:93-97,159-190)
}

func b2i(b bool) C.int32_t {
	if b {
		return 1
	}
	return 0
}

// GlyphSet is rasterised watermark text resident in HBM (ipx_glyphset_*).
type GlyphSet struct {
	x *Context
	c *C.ipx_glyphset
}

func (x *Context) NewGlyphSet(glyphs []Glyph, col [4]uint8) (*GlyphSet, error) {
	arr, free := cGlyphs(glyphs)
	defer free()
	var gs *C.ipx_glyphset
	err := call(func() C.int {
		return C.ipx_glyphset_create(x.c, arr, C.int(len(glyphs)), (*C.uint8_t)(unsafe.Pointer(&col[0])), &gs)
	})
	if err != nil {
		return nil, err
	}
	return &GlyphSet{x, gs}, nil
}

func (g *GlyphSet) Close() { C.ipx_glyphset_destroy(g.x.c, g.c); g.c = nil }

// Plan is the fused resize + thumbnail + watermark pass for frames of one size (ipx_plan_*).
type Plan struct {
	x    *Context
	c    *C.ipx_plan
	Info PlanInfo
	w, h int
}

// PlanInfo holds the actual output sizes after the aspect rules (resize.go:61-75, thumbnail.go:48-65).
type PlanInfo struct {
	ResizeW, ResizeH, ThumbW, ThumbH int
	ThumbCrop                         image.Rectangle
	ResizeBytes, ThumbBytes, WmBytes  int
}

// NewPlan builds the plan for w x h frames.  gs may be nil (watermark = plain copy).
func (x *Context) NewPlan(w, h int, ops Ops, gs *GlyphSet) (*Plan, error) {
	var p C.ipx_plan_params
	p.sw, p.sh = C.int32_t(w), C.int32_t(h)
	if ops.Resize != nil {
		p.do_resize, p.resize_w, p.resize_h, p.keep_aspect = 1, C.int32_t(ops.Resize.W), C.int32_t(ops.Resize.H), b2i(ops.Resize.KeepAspect)
	}
	if ops.Thumb != nil {
		p.do_thumbnail, p.thumb_size, p.crop_to_fit = 1, C.int32_t(ops.Thumb.Size), b2i(ops.Thumb.CropToFit)
	}
	if ops.Watermark {
		p.do_watermark = 1
		if gs != nil {
			p.glyphs = gs.c
		}
	}
	var pl *C.ipx_plan
	if err := call(func() C.int { return C.ipx_plan_create(x.c, &p, &pl) }); err != nil {
		return nil, err
	}
	var i C.ipx_plan_info
	C.ipx_plan_query(pl, &i)
	return &Plan{x: x, c: pl, w: w, h: h, Info: PlanInfo{
		ResizeW: int(i.resize_w), ResizeH: int(i.resize_h), ThumbW: int(i.thumb_w), ThumbH: int(i.thumb_h),
		ThumbCrop:   image.Rect(int(i.thumb_crop.x0), int(i.thumb_crop.y0), int(i.thumb_crop.x1), int(i.thumb_crop.y1)),
		ResizeBytes: int(i.resize_bytes), ThumbBytes: int(i.thumb_bytes), WmBytes: int(i.wm_bytes)}}, nil
}

func (p *Plan) Close() { C.ipx_plan_destroy(p.x.c, p.c); p.c = nil }

func ptr(b []byte) *C.uint8_t {
	if len(b) == 0 {
		return nil
	}
	return (*C.uint8_t)(unsafe.Pointer(&b[0]))
}

// RunHost processes n tightly packed RGBA8 frames (what draw.Draw(rgba, b, img, b.Min, draw.Src) of a decoded *image.RGBA holds) from
// host memory, ideally Pinned: H2D, the fused kernel and D2H of consecutive chunks overlap on the context's lanes.  An output slice
// may be nil to skip that operator's result.  Synchronous.
func (p *Plan) RunHost(n int, src, resizeOut, thumbOut, wmOut []byte) error {
	i := p.Info
	return call(func() C.int {
		return C.ipx_plan_run_host(p.x.c, p.c, C.int(n), ptr(src), C.int(p.w*4), C.size_t(p.w*p.h*4),
			ptr(resizeOut), C.size_t(i.ResizeBytes), ptr(thumbOut), C.size_t(i.ThumbBytes), ptr(wmOut), C.size_t(i.WmBytes))
	})
}

// RunHostYCbCr does the same for decoded JPEGs as image.Decode leaves them (*image.YCbCr planes of n frames, tightly packed per
// plane): per operator the reference converts differently (16-bit per tap inside resize; RGBA8 first for the crop thumbnail and the
// watermark), and the results are those of the reference's helpers on the *image.YCbCr itself.
// The batch descriptor holds pointers into y, cb and cr and is itself passed by pointer: cgo refuses a Go pointer to memory that holds
// unpinned Go pointers, so the planes' bases are pinned for the call (a no-op for Pinned / C memory).  Synchronous: nothing is read
// after the return.
func (p *Plan) RunHostYCbCr(n int, first *image.YCbCr, y, cb, cr, resizeOut, thumbOut, wmOut []byte) error {
	cw, ch := first.CStride, len(first.Cb)/first.CStride
	var pin runtimePinner
	defer pin.Unpin()
	for _, plane := range [][]byte{y, cb, cr} {
		if len(plane) > 0 {
			pin.Pin(&plane[0])
		}
	}
	b := C.ipx_ycbcr_batch{y: ptr(y), cb: ptr(cb), cr: ptr(cr), ystride: C.int32_t(first.YStride), cstride: C.int32_t(cw),
		y_frame_stride: C.size_t(first.YStride * p.h), c_frame_stride: C.size_t(cw * ch), ratio: C.int32_t(first.SubsampleRatio)}
	i := p.Info
	return call(func() C.int {
		return C.ipx_plan_run_host_ycbcr(p.x.c, p.c, C.int(n), &b, ptr(resizeOut), C.size_t(i.ResizeBytes), ptr(thumbOut),
			C.size_t(i.ThumbBytes), ptr(wmOut), C.size_t(i.WmBytes))
	})
}

// Streams are the JPEG objects of one batch; they live in pinned blocks owned by the library until Release.
type Streams struct {
	x                        *Context
	res                      *C.ipx_jpeg_result
	resize, thumb, watermark []C.ipx_bytes
	Status                   []Status // per file, RunJPEGJPEG only: Unsupported = decode this one with Go (progressive, CMYK, another size)
}

func view(b C.ipx_bytes) []byte {
	if b.data == nil {
		return nil
	}
	return unsafe.Slice((*byte)(unsafe.Pointer(b.data)), int(b.len))
}
func (s *Streams) Resize(i int) []byte    { return view(s.resize[i]) }
func (s *Streams) Thumbnail(i int) []byte { return view(s.thumb[i]) }
func (s *Streams) Watermark(i int) []byte { return view(s.watermark[i]) }
func (s *Streams) Release() {
	if s.res != nil {
		C.ipx_jpeg_result_free(s.x.c, s.res)
		s.res = nil
	}
}

// RunHostJPEG: frames in, operators and jpeg.Encode(q) on the GPU (resize.go:80, thumbnail.go:70, watermark.go:68), only the finished
// streams cross the link.  The bytes are the ones Go's image/jpeg writer produces for the same *image.RGBA.
func (p *Plan) RunHostJPEG(n int, src []byte, quality int) (*Streams, error) {
	s := &Streams{x: p.x, resize: make([]C.ipx_bytes, n), thumb: make([]C.ipx_bytes, n), watermark: make([]C.ipx_bytes, n)}
	err := call(func() C.int {
		return C.ipx_plan_run_host_jpeg(p.x.c, p.c, C.int(n), ptr(src), C.int(p.w*4), C.size_t(p.w*p.h*4), C.int(quality),
			&s.resize[0], &s.thumb[0], &s.watermark[0], &s.res)
	})
	if err != nil {
		return nil, err
	}
	return s, nil
}

// RunJPEGJPEG: the objects as fileRepo.GetOriginal returned them in, three objects per file out -- image.Decode (image_processor.go:47),
// every operator and jpeg.Encode on the GPU.  files must be C or pinned memory for the duration of the call (pass Pinned slices, or
// copy with C.CBytes); Status[i] != OK marks the files Go has to process itself.
func (p *Plan) RunJPEGJPEG(files [][]byte, quality int) (*Streams, error) {
	n := len(files)
	cf := (*[1 << 24]C.ipx_bytes)(C.malloc(C.size_t(n) * C.size_t(unsafe.Sizeof(C.ipx_bytes{}))))
	defer C.free(unsafe.Pointer(cf))
	var pin runtimePinner
	defer pin.Unpin()
	for i, f := range files {
		pin.Pin(&f[0])
		cf[i] = C.ipx_bytes{data: (*C.uint8_t)(unsafe.Pointer(&f[0])), len: C.size_t(len(f))}
	}
	s := &Streams{x: p.x, resize: make([]C.ipx_bytes, n), thumb: make([]C.ipx_bytes, n), watermark: make([]C.ipx_bytes, n)}
	st := make([]C.int, n)
	err := call(func() C.int {
		return C.ipx_plan_run_jpeg_jpeg(p.x.c, p.c, C.int(n), &cf[0], C.int(quality), &s.resize[0], &s.thumb[0], &s.watermark[0], &st[0], &s.res)
	})
	if err != nil {
		return nil, err
	}
	s.Status = make([]Status, n)
	for i := range st {
		s.Status[i] = Status(st[i])
	}
	return s, nil
}
