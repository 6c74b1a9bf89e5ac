// TEST SCAFFOLDING: see stub_tesseract.h
#pragma once
#include <tesseract/stub_tesseract.h>
