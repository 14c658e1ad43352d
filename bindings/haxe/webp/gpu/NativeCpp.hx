package webp.gpu;

#if cpp
/**
    hxcpp externs: the entry points of include/webp_amd.h bound directly (`@:native`), the library linked through
    `@:buildXml`.  Pointers come from `cpp.NativeArray.address(bytes.getData(), 0).raw` (see Gpu.hx).
**/
@:include("webp_amd.h")
@:buildXml('
<target id="haxe">
  <lib name="-lwebp_amd"/>
  <flag value="-L${WEBP_AMD_LIB_DIR}"/>
</target>
<files id="haxe">
  <compilerflag value="-I${WEBP_AMD_INCLUDE_DIR}"/>
</files>')
extern class NativeCpp {
    @:native("wp_init") static function init(device:Int):Int;
    @:native("wp_shutdown") static function shutdown():Int;
    @:native("wp_device_count") static function deviceCount():Int;
    @:native("wp_last_error") static function lastError(buf:cpp.RawPointer<cpp.Char>, cap:Int):Int;
    @:native("wp_vp8_decode_to_argb") static function vp8DecodeToArgb(n:Int, frames:cpp.RawConstPointer<cpp.Void>,
        mb:cpp.RawConstPointer<cpp.UInt32>, nMb:cpp.SizeT, filt:cpp.RawConstPointer<cpp.UInt32>,
        coeff:cpp.RawConstPointer<cpp.Int16>, nBlocks:cpp.SizeT,
        y:cpp.RawPointer<cpp.UInt8>, u:cpp.RawPointer<cpp.UInt8>, v:cpp.RawPointer<cpp.UInt8>,
        alpha:cpp.RawConstPointer<cpp.UInt8>, argb:cpp.RawPointer<cpp.UInt8>):Int;
    @:native("wp_vp8_recon") static function vp8Recon(n:Int, frames:cpp.RawConstPointer<cpp.Void>, mb:cpp.RawConstPointer<cpp.UInt32>, nMb:cpp.SizeT,
        coeff:cpp.RawConstPointer<cpp.Int16>, nBlocks:cpp.SizeT, y:cpp.RawPointer<cpp.UInt8>, u:cpp.RawPointer<cpp.UInt8>, v:cpp.RawPointer<cpp.UInt8>):Int;
    @:native("wp_vp8_loopfilter") static function vp8LoopFilter(n:Int, frames:cpp.RawConstPointer<cpp.Void>, filt:cpp.RawConstPointer<cpp.UInt32>, nMb:cpp.SizeT,
        y:cpp.RawPointer<cpp.UInt8>, u:cpp.RawPointer<cpp.UInt8>, v:cpp.RawPointer<cpp.UInt8>):Int;
    @:native("wp_yuv420_to_argb") static function yuv420ToArgb(n:Int, frames:cpp.RawConstPointer<cpp.Void>, y:cpp.RawConstPointer<cpp.UInt8>,
        u:cpp.RawConstPointer<cpp.UInt8>, v:cpp.RawConstPointer<cpp.UInt8>, alpha:cpp.RawConstPointer<cpp.UInt8>, argb:cpp.RawPointer<cpp.UInt8>):Int;
    @:native("wp_vp8l_inverse") static function vp8lInverse(n:Int, frames:cpp.RawConstPointer<cpp.Void>, pix:cpp.RawConstPointer<cpp.UInt8>, pixLen:cpp.SizeT,
        aux:cpp.RawConstPointer<cpp.UInt8>, auxLen:cpp.SizeT, out:cpp.RawPointer<cpp.UInt8>, outLen:cpp.SizeT, swizzle:Int):Int;
    @:native("wp_alpha_unfilter") static function alphaUnfilter(n:Int, frames:cpp.RawConstPointer<cpp.Void>, src:cpp.RawConstPointer<cpp.UInt8>, srcLen:cpp.SizeT,
        alpha:cpp.RawPointer<cpp.UInt8>, alphaLen:cpp.SizeT):Int;
    @:native("wp_vp8l_resolve") static function vp8lResolve(n:Int, frames:cpp.RawConstPointer<cpp.Void>, tokens:cpp.RawConstPointer<cpp.UInt32>, nWords:cpp.SizeT,
        pix:cpp.RawPointer<cpp.UInt8>, pixLen:cpp.SizeT):Int;
    @:native("wp_anim_compose") static function animCompose(n:Int, frames:cpp.RawConstPointer<cpp.Void>, canvasW:Int, canvasH:Int,
        argb:cpp.RawConstPointer<cpp.UInt8>, argbLen:cpp.SizeT, canvases:cpp.RawPointer<cpp.UInt8>, canvasesLen:cpp.SizeT):Int;
}
#end
