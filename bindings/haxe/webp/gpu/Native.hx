package webp.gpu;

#if hl
/**
    HashLink externs over webpgpu.hdll (bindings/hashlink/webpgpu_hl.c), one per entry point of include/webp_amd.h
    that the Haxe host side calls.  All return the library's status (0 ok, < 0 error).
**/
@:hlNative("webpgpu")
extern class Native {
    static function init(device:Int):Int;
    static function shutdown():Int;
    static function device_count():Int;
    static function last_error(buf:hl.Bytes, cap:Int):Int;
    static function vp8_decode_to_argb(n:Int, frames:hl.Bytes, mb:hl.Bytes, nMb:Int, filt:hl.Bytes, coeff:hl.Bytes, nBlocks:Int,
                                       y:hl.Bytes, u:hl.Bytes, v:hl.Bytes, alpha:hl.Bytes, argb:hl.Bytes):Int;
    static function vp8_recon(n:Int, frames:hl.Bytes, mb:hl.Bytes, nMb:Int, coeff:hl.Bytes, nBlocks:Int, y:hl.Bytes, u:hl.Bytes, v:hl.Bytes):Int;
    static function vp8_loopfilter(n:Int, frames:hl.Bytes, filt:hl.Bytes, nMb:Int, y:hl.Bytes, u:hl.Bytes, v:hl.Bytes):Int;
    static function yuv420_to_argb(n:Int, frames:hl.Bytes, y:hl.Bytes, u:hl.Bytes, v:hl.Bytes, alpha:hl.Bytes, argb:hl.Bytes):Int;
    static function vp8l_inverse(n:Int, frames:hl.Bytes, pix:hl.Bytes, pixLen:Int, aux:hl.Bytes, auxLen:Int, out:hl.Bytes, outLen:Int, swizzle:Int):Int;
    static function alpha_unfilter(n:Int, frames:hl.Bytes, src:hl.Bytes, srcLen:Int, alpha:hl.Bytes, alphaLen:Int):Int;
    static function vp8l_resolve(n:Int, frames:hl.Bytes, tokens:hl.Bytes, nWords:Int, pix:hl.Bytes, pixLen:Int):Int;
    static function anim_compose(n:Int, frames:hl.Bytes, canvasW:Int, canvasH:Int, argb:hl.Bytes, argbLen:Int, canvases:hl.Bytes, canvasesLen:Int):Int;
}
#end
