module ipx/gen_go_vectors

go 1.24

// the versions the reference pins (go.mod:42,45 of sj-shoff/ImageProcessor)
require (
	github.com/golang/freetype v0.0.0-20170609003504-e2365dfdc4a0
	golang.org/x/image v0.33.0
)
