// Package ipx binds libipx, the MI355X pixel worker, into the reference's Go worker (sj-shoff/ImageProcessor).
//
// It replaces, behind the reference's own seams, the per-pixel work of internal/usecase/processor: resizeImage
// (operations/resize.go:121-125), cropAndResize (operations/thumbnail.go:114-132), addTextWatermark (operations/watermark.go:86-157)
// and, batched, (*ImageProcessor).Process (image_processor.go:39-102) together with image.Decode and jpeg.Encode for JPEG objects.
//
// SOURCE ONLY: the build image of the libipx repository has no Go toolchain, so this package has never been compiled there.  The C ABI it
// binds (include/ipx.h) is exercised from C (tests/c_abi_consumer.c), C++ and Python.  Build with CGO_ENABLED=1 (the reference's
// dockerfile:12-13 sets 0) and ship libipx.so + /opt/rocm/lib in the worker image.
package ipx

/*
#cgo CFLAGS: -I${SRCDIR}/../../include
#cgo LDFLAGS: -L${SRCDIR}/../../imageprocessor_amd -lipx -L/opt/rocm/lib -lamdhip64
#include <stdlib.h>
#include "ipx.h"
*/
import "C"

import (
	"errors"
	"fmt"
	"image"
	"runtime"
	"unsafe"
)

// Status mirrors ipx_status; Unsupported means "valid in the reference but outside the GPU path": run the CPU operator.
type Status int

const (
	OK          Status = 0
	Invalid     Status = -1
	NoMem       Status = -2
	HIPFailure  Status = -3
	Unsupported Status = -4
	NoDevice    Status = -5
)

// Error carries the status and the library's text (ipx_last_error, thread-local: read on the calling OS thread).
type Error struct {
	Status Status
	Text   string
}

func (e *Error) Error() string { return fmt.Sprintf("ipx: %s (status %d)", e.Text, int(e.Status)) }

// call runs f on a locked OS thread and turns a negative status into an *Error with the text of that thread.
func call(f func() C.int) error {
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	if rc := f(); rc < 0 {
		return &Error{Status(rc), C.GoString(C.ipx_last_error())}
	}
	return nil
}

// IsUnsupported reports whether err says "keep the CPU path for this one" (progressive JPEG, a 2 GiB frame, ...).
func IsUnsupported(err error) bool {
	var e *Error
	return errors.As(err, &e) && e.Status == Unsupported
}

// Context owns the GPU state of one worker process on one device.  Created next to processor.NewImageProcessor
// (image_processor.go:29); every method is safe to call from any goroutine.
type Context struct{ c *C.ipx_ctx }

// New opens device `device` (-1: IPX_DEVICE / LOCAL_RANK / 0) with `lanes` staging lanes (0: the library's default of 4).
func New(device, lanes int) (*Context, error) {
	cfg := C.ipx_config{device: C.int32_t(device), lanes: C.int32_t(lanes)}
	var c *C.ipx_ctx
	if err := call(func() C.int { return C.ipx_create(&cfg, &c) }); err != nil {
		return nil, err
	}
	return &Context{c}, nil
}

func (x *Context) Close() { C.ipx_destroy(x.c); x.c = nil }

// DeviceCount is the number of gfx950 devices visible to the process.
func DeviceCount() int { return int(C.ipx_device_count()) }

// FrameSupported says whether a w x h frame with this row stride can take the GPU path at all (ipx_frame_supported: frames of
// 2 GiB or more, or with a side beyond 65535, stay on the CPU).
func FrameSupported(w, h, stride, bytesPerPixel int) bool {
	return C.ipx_frame_supported(C.int(w), C.int(h), C.longlong(stride), C.int(bytesPerPixel)) == 0
}

func rect(r image.Rectangle) C.ipx_rect {
	return C.ipx_rect{x0: C.int32_t(r.Min.X), y0: C.int32_t(r.Min.Y), x1: C.int32_t(r.Max.X), y1: C.int32_t(r.Max.Y)}
}

func pix(p []uint8) *C.uint8_t { return (*C.uint8_t)(unsafe.Pointer(&p[0])) }

// ScaleBilinear is xdraw.BiLinear.Scale(dst, dr, src, sr, op, nil) for *image.RGBA <- *image.RGBA: the body of resizeImage
// (operations/resize.go:121-125) and of cropAndResize (thumbnail.go:128-131, with sr = the crop rectangle).  Go slices may be
// passed to this synchronous call (cgo pins them for its duration; the library keeps no pointer).
func (x *Context) ScaleBilinear(dst *image.RGBA, dr image.Rectangle, src *image.RGBA, sr image.Rectangle, over bool) error {
	op := C.int(C.IPX_OP_SRC)
	if over {
		op = C.IPX_OP_OVER
	}
	return call(func() C.int {
		return C.ipx_scale_bilinear_rgba8(x.c,
			pix(dst.Pix), C.int(dst.Rect.Dx()), C.int(dst.Rect.Dy()), C.int(dst.Stride), rect(dr.Sub(dst.Rect.Min)),
			pix(src.Pix), C.int(src.Rect.Dx()), C.int(src.Rect.Dy()), C.int(src.Stride), rect(sr.Sub(src.Rect.Min)), op)
	})
}

// Glyph is one draw.DrawMask call of freetype.Context.DrawString (watermark.go:151): the A8 mask, the destination rectangle and the
// mask point aligned with Dr.Min (DrawString passes mp = image.Point{0, dr.Min.Y - glyphRect.Min.Y}: reproduce exactly that).
type Glyph struct {
	Mask *image.Alpha
	Dr   image.Rectangle
	Mp   image.Point
}

// cGlyphs lays the glyph list out in C memory (a Go pointer to Go pointers must not cross the boundary) and pins the masks.
func cGlyphs(glyphs []Glyph) (*C.ipx_glyph, func()) {
	if len(glyphs) == 0 {
		return nil, func() {}
	}
	arr := (*[1 << 20]C.ipx_glyph)(C.malloc(C.size_t(len(glyphs)) * C.size_t(unsafe.Sizeof(C.ipx_glyph{}))))
	var pin runtime.Pinner
	for i, g := range glyphs {
		b := g.Mask.Bounds()
		if len(g.Mask.Pix) > 0 {
			pin.Pin(&g.Mask.Pix[0])
			arr[i].mask = pix(g.Mask.Pix)
		}
		arr[i].mw, arr[i].mh, arr[i].mstride = C.int32_t(b.Dx()), C.int32_t(b.Dy()), C.int32_t(g.Mask.Stride)
		arr[i].dr = rect(g.Dr)
		arr[i].mpx, arr[i].mpy = C.int32_t(g.Mp.X-b.Min.X), C.int32_t(g.Mp.Y-b.Min.Y)
	}
	return &arr[0], func() { pin.Unpin(); C.free(unsafe.Pointer(arr)) }
}

// CompositeGlyphs is the compositing half of DrawString (draw.DrawMask per glyph, in order, Over), in place on dst.
func (x *Context) CompositeGlyphs(dst *image.RGBA, glyphs []Glyph, col [4]uint8) error {
	if len(glyphs) == 0 {
		return nil
	}
	arr, free := cGlyphs(glyphs)
	defer free()
	return call(func() C.int {
		return C.ipx_composite_glyphs_rgba8(x.c, pix(dst.Pix), C.int(dst.Rect.Dx()), C.int(dst.Rect.Dy()), C.int(dst.Stride),
			arr, C.int(len(glyphs)), (*C.uint8_t)(unsafe.Pointer(&col[0])))
	})
}

// Pinned is hipHostMalloc'd staging handed out as a Go slice: what the batched and asynchronous entries want to read from and write to.
// The slice must not be used after Free; C owns the memory.
type Pinned struct {
	Bytes []byte
	free  func()
}

func (p *Pinned) Free() { p.free(); p.Bytes = nil }

func (x *Context) Pinned(n int) (*Pinned, error) {
	p := C.ipx_host_alloc(x.c, C.size_t(n))
	if p == nil {
		return nil, &Error{NoMem, C.GoString(C.ipx_last_error())}
	}
	return &Pinned{unsafe.Slice((*byte)(p), n), func() { C.ipx_host_free(x.c, p) }}, nil
}
